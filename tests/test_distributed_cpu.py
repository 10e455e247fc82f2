"""world_size-2 tests (CPU): the sharding / broadcast / gather logic of the N > 1 path over
both host transports -- the package's own TCP rendezvous (what the launcher's environment
selects) and a torch.distributed gloo group wrapped as a rendezvous.Exchange.  The per-shard
compute is injected (the oracle stands in for the device call, which needs a GPU); equality
with the unsharded result proves the decomposition is exact."""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_exchange(rank, world):
    """torch.distributed (gloo) as the host transport: lives in the TEST, the package has no torch."""
    import torch.distributed as dist
    from dsptoolbox_amd.rendezvous import Exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Gloo(Exchange):
        def __init__(self):
            self.rank, self.world = rank, world

        def broadcast_bytes(self, data, src=0):
            box = [data if rank == src else None]
            dist.broadcast_object_list(box, src=src)
            return box[0]

        def allgather_bytes(self, data):
            parts = [None] * world
            dist.all_gather_object(parts, data)
            return parts

        def barrier(self):
            dist.barrier()

        def close(self):
            dist.destroy_process_group()

    return Gloo()


def _worker(rank, world, port, q, transport="tcp"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    from dsptoolbox_amd import distributed as dd
    from oracle import dsp_oracle as orc
    if transport == "gloo":
        dd.init(_gloo_exchange(rank, world))
    else:
        dd.init()  # rendezvous.from_environment: TCP star on MASTER_PORT + 23
    assert dd.world() == (rank, world)
    assert "torch" not in sys.modules or transport == "gloo"
    try:
        rng = np.random.default_rng(42)
        n, n_cy = 6000, 5  # 5 channels over 2 ranks: 3 + 2
        x = rng.standard_normal((n, 1)) * 0.3
        y = np.stack([np.convolve(x[:, 0], rng.standard_normal(8))[:n] for _ in range(n_cy)], axis=1)
        y += 0.01 * rng.standard_normal(y.shape)
        # only rank 0 "owns" the sweep: broadcast it
        xb = dd.broadcast_array(x if rank == 0 else None, src=0)
        assert np.array_equal(xb, x)

        def compute(ys, xs, fs, W, mode, **kw):
            return orc.compute_transfer_function(ys, xs, fs, W, mode, **kw)

        tf, coh = dd.welch_transfer_function_sharded(y, xb, 48000, 256, "H1", compute=compute,
                                                     detrend=True)
        rt, rc = orc.compute_transfer_function(y, x, 48000, 256, "H1", detrend=True)
        ok = (np.array_equal(tf[1:], rt[1:]) and np.array_equal(coh[1:], rc[1:])
              and tf.shape == (129, n_cy))
        # per-channel input is sharded with the output
        x5 = np.repeat(x, n_cy, axis=1) * np.arange(1, n_cy + 1)
        tf2, _ = dd.welch_transfer_function_sharded(y, x5, 48000, 256, "H2", compute=compute)
        rt2, _ = orc.compute_transfer_function(y, x5, 48000, 256, "H2")
        ok = ok and np.array_equal(tf2[1:], rt2[1:])
        # more ranks than units: empty shards
        tf3, _ = dd.welch_transfer_function_sharded(y[:, :1], xb, 48000, 256, "H1", compute=compute)
        ok = ok and tf3.shape == (129, 1)
        # FIR bank: bands sharded (Parallel), channels sharded (Summed); batched deconvolution:
        # items sharded -- the oracle stands in for the device call
        from dsptoolbox_amd import backend
        taps = [rng.standard_normal(31) for _ in range(5)]
        xs = rng.standard_normal((2000, 3))

        def fir(xx, tt, mode):
            name = {backend.DS_FB_PARALLEL: "Parallel", backend.DS_FB_SUMMED: "Summed"}[mode]
            r = orc.filterbank_fir(list(tt), xx, name)
            return np.transpose(r, (2, 0, 1)) if name == "Parallel" else r

        yp = dd.fir_filter_bank_sharded(xs, taps, backend.DS_FB_PARALLEL, compute=fir)
        ok = ok and np.array_equal(yp, fir(xs, taps, backend.DS_FB_PARALLEL)) and yp.shape == (5, 2000, 3)
        ysum = dd.fir_filter_bank_sharded(xs, taps, backend.DS_FB_SUMMED, compute=fir)
        ok = ok and np.array_equal(ysum, fir(xs, taps, backend.DS_FB_SUMMED))
        items = rng.standard_normal((7, 256, 2))
        rinv = rng.standard_normal(129) + 1j * rng.standard_normal(129)

        def div(it, nfft, r, n_out):
            return np.fft.irfft(np.fft.rfft(it, n=nfft, axis=1) * r[None, :, None], n=nfft, axis=1)[:, :n_out]

        d = dd.spectral_division_sharded(items, 256, rinv, 200, compute=div)
        ok = ok and np.array_equal(d, div(items, 256, rinv, 200)) and d.shape == (7, 200, 2)
        # CSM: frequency bins sharded
        def csm_bins(td, fs, W, b0, b1, **kw):
            return orc.csm_welch_batched(td, fs, W, "hann", 50, True, "FFTBackward")[1][b0:b1]

        fcs, cs = dd.csm_welch_sharded(xs, 48000, 128, compute=csm_bins)
        ok = ok and cs.shape == (65, 3, 3) and np.array_equal(cs, csm_bins(xs, 48000, 128, 0, 65))
        q.put((rank, bool(ok), dd.shard_range(n_cy, world, rank)))
    finally:
        dd.shutdown()


def test_shard_range_properties():
    from dsptoolbox_amd.distributed import shard_range
    for n in (0, 1, 7, 64, 1024, 1025):
        for ws in (1, 2, 3, 8):
            spans = [shard_range(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(64, 8, 3) == (24, 32) and shard_range(1024, 8, 7) == (896, 1024)


def test_package_imports_no_torch():
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import dsptoolbox_amd, dsptoolbox_amd.distributed, "
            "dsptoolbox_amd.rendezvous; assert 'torch' not in sys.modules" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True, timeout=120)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("transport", ["tcp", "gloo"])
def test_sharded_transfer_function_world2(transport):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, transport)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert {r[0]: r[2] for r in res} == {0: (0, 3), 1: (3, 5)}


def test_host_array_wire_format_is_data_only():
    """the header of a host array is a fixed struct (no eval); malformed headers raise."""
    import struct
    from dsptoolbox_amd import distributed as dd
    rng = np.random.default_rng(0)
    for a in (rng.standard_normal((3, 0, 2)), rng.standard_normal(7).astype(np.float32),
              (rng.standard_normal((2, 3)) + 1j).astype(np.complex64), np.arange(6).reshape(2, 3),
              np.array(3.5), np.zeros((4, 2), dtype=bool)):
        b = dd._unpack(dd._pack(a))
        assert b.dtype == a.dtype and b.shape == a.shape and np.array_equal(a, b)
    good = dd._pack(np.arange(4.0))
    # the round-2 format (repr of a tuple) and other text is refused, nothing is evaluated
    evil = b"(__import__('os').system('true'), '<f8')"
    with pytest.raises(ValueError):
        dd._unpack(struct.pack("<I", len(evil)) + evil)
    with pytest.raises(ValueError):
        dd._unpack(good[:-1])  # payload shorter than the header says
    with pytest.raises(ValueError):
        dd._unpack(good[:4] + bytes([200, 1]) + good[6:])  # unknown dtype code
    with pytest.raises(ValueError):
        dd._unpack(good[:6] + struct.pack("<q", -4) + good[14:])
    with pytest.raises(TypeError):
        dd._pack(np.array([object()]))


def _handshake_server(port, q):
    sys.path.insert(0, ROOT)
    from dsptoolbox_amd.rendezvous import TcpExchange
    ex = TcpExchange(0, 2, "127.0.0.1", port, timeout_s=30.0, token=b"k" * 32)
    q.put(ex.allgather_bytes(b"zero"))
    ex.close()


@pytest.mark.timeout(90)
def test_tcp_rendezvous_rejects_strangers():
    """a connection without the job's token, with a rank out of range or with rank 0 is
    dropped; the real rank is still admitted afterwards."""
    import hashlib
    import hmac
    import multiprocessing as mp
    import struct
    import time
    from dsptoolbox_amd.rendezvous import TcpExchange
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    srv = ctx.Process(target=_handshake_server, args=(port, q))
    srv.start()

    def knock(rank, token):
        for _ in range(200):
            try:
                s = socket.create_connection(("127.0.0.1", port), timeout=5.0)
                break
            except OSError:
                time.sleep(0.05)
        s.settimeout(10.0)
        ch = s.recv(16)
        me = struct.pack("<I", rank)
        s.sendall(me + hmac.new(token, ch + me, hashlib.sha256).digest() + os.urandom(16))
        try:
            ans = s.recv(1)
        except OSError:
            ans = b""
        s.close()
        return ans

    assert knock(1, b"wrong" * 8) == b""      # no token
    assert knock(0, b"k" * 32) == b""         # rank 0 is the server itself
    assert knock(7, b"k" * 32) == b""         # outside the world
    ex = TcpExchange(1, 2, "127.0.0.1", port, timeout_s=30.0, token=b"k" * 32)
    assert ex.allgather_bytes(b"one") == [b"zero", b"one"]
    ex.close()
    assert q.get(timeout=30) == [b"zero", b"one"]
    srv.join(timeout=30)
    assert srv.exitcode == 0


def test_tcp_rendezvous_is_mutual_and_needs_a_secret_off_loopback(monkeypatch):
    """rank 0 proves itself to the peers too (a listener that does not know the job's token is
    refused by the connecting rank), and an outside interface is only used with an explicit shared secret."""
    import threading
    from dsptoolbox_amd.rendezvous import TcpExchange
    port = _free_port()
    srv = socket.socket()
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", port))
    srv.listen(1)

    def impostor():  # accepts anybody, answers "ok" with a made-up proof
        conn, _ = srv.accept()
        conn.sendall(os.urandom(16))
        conn.recv(4 + 32 + 16)
        conn.sendall(b"\x01" + os.urandom(32))
        conn.close()

    th = threading.Thread(target=impostor, daemon=True)
    th.start()
    with pytest.raises(ConnectionError, match="does not know this job's token"):
        TcpExchange(1, 2, "127.0.0.1", port, timeout_s=10.0, token=b"k" * 32)
    th.join(timeout=10)
    srv.close()
    monkeypatch.delenv("DSPTOOLBOX_AMD_RDZV_TOKEN", raising=False)
    with pytest.raises(RuntimeError, match="DSPTOOLBOX_AMD_RDZV_TOKEN"):
        TcpExchange(1, 2, "10.1.2.3", port, timeout_s=1.0)


_BENCH_DIST_RANK = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import bench
d = bench.Dist(2)
import numpy as np
class Ctx:
    synced = 0
    handle = 1
    uploaded = None
    def sync(self):
        Ctx.synced += 1
    def malloc(self, n):
        return 4096
    def free(self, p):
        pass
    def upload(self, ptr, arr):
        Ctx.uploaded = np.array(arr, copy=True)
ctx = Ctx()
# the shared FIR taps / inverse sweep without RCCL (ranks sharing a GPU, BENCH_BCAST=host): rank 0 owns the values,
# the host exchange carries them, every rank uploads what rank 0 holds
taps = np.arange(12, dtype=np.float32).reshape(3, 4) * (1.0 if d.rank == 0 else -7.0)
buf, bcast_ms = bench.shared_upload(ctx, d, False, taps)
shared_ok = bcast_ms is None and np.array_equal(Ctx.uploaded, np.arange(12, dtype=np.float32).reshape(3, 4))
d.barrier_sync(ctx)
m = d.max_over_ranks(1.5 + d.rank)
some_ok = d.all_ok(d.rank == 0)      # one rank says no: every rank must hear no
all_ok = d.all_ok(True)
ident = d.bcast_bytes(bytes(range(128)) if d.rank == 0 else b"", 128)
d.barrier_sync(ctx)
print(json.dumps(dict(rank=d.rank, world=d.world, max=m, some_ok=some_ok, all_ok=all_ok, ident=ident == bytes(range(128)),
                      synced=Ctx.synced, torch="torch" in sys.modules, shared_ok=bool(shared_ok))), flush=True)
d.finish()
"""


@pytest.mark.timeout(120)
def test_bench_rank_plumbing_without_torch_world2():
    """bench.py's rendezvous / barrier / max over ranks / ok-reduction / id broadcast ride on the
    package's own host exchange; two ranks as the launcher's environment would make them, no GPU needed for this
    part, and torch is never imported."""
    import json
    import subprocess
    port = _free_port()
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    procs = []
    for r in range(2):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", _BENCH_DIST_RANK.format(root=ROOT)], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=100) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    got = sorted((json.loads(so.strip().splitlines()[-1]) for so, _ in outs), key=lambda d: d["rank"])
    assert [g["rank"] for g in got] == [0, 1] and all(g["world"] == 2 for g in got)
    for g in got:
        assert g["max"] == 2.5 and g["some_ok"] is False and g["all_ok"] is True and g["ident"] and g["synced"] == 2
        assert g["torch"] is False
        assert g["shared_ok"] is True  # bench.shared_upload's host-exchange fallback (fir_bank taps, deconv inverse sweep)


def test_bench_refuses_a_missing_exchange():
    """A rank whose peers never arrive cannot be timed: exit status 3, never a JSON line."""
    import subprocess
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")},
               RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               DSPTOOLBOX_AMD_RDZV_TIMEOUT="2")
    code = ("import sys; sys.path.insert(0, %r); import bench; bench.Dist(2)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=100)
    assert r.returncode == 3 and "host exchange not available" in r.stderr and "{" not in r.stdout, (r.returncode, r.stderr[-500:])


@pytest.mark.timeout(300)
def test_bench_gpus_without_launcher_never_reports_one_gpu():
    """`python bench.py --gpus N` with no launcher environment starts N ranks itself
    (children of a process that has not touched the GPU; since round 4 without torch.distributed.run); a launcher
    environment of another size is refused.  Here (no GPU) the children fail -- what matters: no JSON line with n_gpus 1 is
    ever printed for --gpus 2, and the exit status is non-zero."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert "starting 2 ranks" in r.stderr, r.stderr[-2000:]
    assert '"n_gpus": 1' not in r.stdout
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
    # a launcher that made ONE rank while --gpus says 2: refused before anything is measured
    env1 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r1 = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                        capture_output=True, text=True, timeout=200, env=env1, cwd=ROOT)
    assert r1.returncode == 2 and "refusing" in r1.stderr and '"n_gpus"' not in r1.stdout, (r1.returncode, r1.stderr[-1000:])
