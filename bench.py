#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X spectral hot path.

    python bench.py --gpus N --steps K --warmup W [--workload welch_h1|welch_h1_1024|fir_bank|csm|deconv]

Default workload (BASELINE.json configs[1], the one the metric is quoted on):
64-channel Welch H1 transfer-function estimation, one sweep input channel,
2^20 samples per channel, nfft 4096, Hann, 50 % overlap.  One step = one
ds_welch_tf_dev call over inputs that are already resident in HBM.
N > 1: one process per GPU, as the driver launches it (`python -m torch.distributed.run ... bench.py --gpus N`:
torch is the process launcher and nothing else) or, with no launcher environment, started by this script
itself.  The ranks meet over the package's own host exchange (dsptoolbox_amd/rendezvous.py: barrier, max over
ranks, the 128-byte RCCL id) and the ONLY RCCL communicator of the job is the library's (ds_comm_init); the line
reports how many ranks RCCL itself counts ("rccl_ranks", ncclCommCount through ds_comm_count).
  --scaling weak   (default) every rank owns an independent 64-channel batch, no data-path
                   collective; the shared sweep channel is broadcast once over RCCL/xGMI
                   before the timed region;
  --scaling strong the ONE 64-channel job is split: output channels /
                   bands / items / bins by shard_range, the shared input by ds_bcast, and every
                   step ends with the RCCL all-gather of the result slices (ds_allgather).
At N = 1, `--predict-ranks 8` also times the per-rank shard shape of an 8-GPU strong-scaling
job and prints the implied speed-up T(full) / T(shard) ("shard_prediction"; profiles/rNN_shard_prediction.jsonl).
At N = 1 the default line also carries "workloads": the other BASELINE configs (welch_h1_1024, fir_bank, csm,
deconv), each timed the same way over a shorter region with a bounded CPU leg (`--no-workloads` skips them).

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md

FS = 48000

# bench kernel name (ds_profile_*) -> how rocprofv3's kernel trace names the same kernel
KERNEL_HINTS = {"welch4096_main": ("welch4096::k_y3<", "welch4096::k_y<"),
                "welch1024_main": ("welch1k::k_y<",), "welch_yacc": ("k_yacc",),
                "fir": ("fir4k::k_fir3<", "fir4k::k_fir<", "fir16k::k_fir<true>", "fir16k::k_fir", "k_fir<"),
                "csm_gemm": ("k_csm_fused", "k_csm_gemm"), "stft": ("k_stft_wave", "k_stft"),
                "deconv": ("k_deconv_p", "k_deconv3", "k_deconv")}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # timed region of tens of ms
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--predict-ranks", type=int, default=0,
                    help="N = 1: also time the per-rank shard of a strong-scaling job over this many GPUs and "
                         "print shard_prediction (off by default: those launches reuse the dominant kernel with a "
                         "smaller grid and would dilute its average in a rocprofv3 trace of this command)")
    ap.add_argument("--workload", default="welch_h1",
                    choices=["welch_h1", "welch_h1_1024", "fir_bank", "csm", "deconv"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-channels", type=int, default=64,
                    help="output channels of the bounded CPU-baseline sample")
    ap.add_argument("--detrend", type=int, default=1)
    ap.add_argument("--no-workloads", action="store_true",
                    help="default workload at N = 1: do not time the other BASELINE configs into \"workloads\"")
    ap.add_argument("--preheat-ms", type=float, default=40.0,
                    help="untimed steps run for this long before the warm-up (clock ramp of an idle chip; 0 = none)")
    ap.add_argument("--steady-steps", type=int, default=2000,
                    help="steps of the second timed region behind the K steps of the command line -> \"steady_state\" (0 = none)")
    ap.add_argument("--workload-steps", type=int, default=200,
                    help="timed steps of each entry of \"workloads\" (40 until round 4: 5 ms of timed region sits inside the clock ramp of an idle chip -- the 1024-sample Welch step read 0.130 ms there and 0.103 ms over 60 000 steps)")
    return ap.parse_args()


# ---------------------------------------------------------------------------
class Dist:
    """Rendezvous, barrier, max over ranks: the package's own host exchange (a TCP star around rank 0 at
    MASTER_ADDR : MASTER_PORT + 23 of the launcher's environment, dsptoolbox_amd/rendezvous.py).  No torch:
    `torch.distributed.run` may have started the ranks, this process never imports it."""

    def __init__(self, n_gpus: int):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # ranks that share ONE GPU (a rehearsal on a one-GPU box): RCCL refuses two ranks on a device
        self.rehearsal = os.environ.get("BENCH_DIST_BACKEND", "") == "gloo" or os.environ.get("BENCH_BCAST", "rccl") != "rccl"
        self.ex = None
        if self.world != n_gpus:
            # (main() starts the ranks itself when there is no launcher environment at all; a launcher
            # that made a different number of ranks than --gpus says is an error, never an N = 1 line)
            print(f"[bench] --gpus {n_gpus} but the launcher environment says WORLD_SIZE={self.world}; "
                  "refusing to report a line for the wrong number of GPUs", file=sys.stderr, flush=True)
            sys.exit(2)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            from dsptoolbox_amd import distributed as dd
            try:
                self.ex = dd.init()
            except Exception as ex:  # noqa: BLE001 - e.g. the exchange's port is taken, a rank never arrived
                print(f"[bench] rank {self.rank}/{self.world}: host exchange not available ({ex!r}); the ranks cannot "
                      "be timed together. Exiting with status 3.", file=sys.stderr, flush=True)
                sys.exit(3)

    def barrier_sync(self, ctx):
        ctx.sync()
        if self.ex is not None:
            self.ex.barrier()

    def _gather_f64(self, v: float):
        import struct
        return [struct.unpack("<d", b)[0] for b in self.ex.allgather_bytes(struct.pack("<d", float(v)))]

    def max_over_ranks(self, v: float) -> float:
        return v if self.ex is None else max(self._gather_f64(v))

    def all_ok(self, ok: bool) -> bool:
        return ok if self.ex is None else min(self._gather_f64(1.0 if ok else 0.0)) > 0.5

    def bcast_bytes(self, b: bytes, n: int) -> bytes:
        return b if self.ex is None else self.ex.broadcast_bytes(b if self.rank == 0 else None, src=0)

    def finish(self):
        if self.ex is not None:
            try:
                self.ex.barrier()
            except Exception:  # pragma: no cover - teardown only
                pass
            from dsptoolbox_amd import distributed as dd
            dd.shutdown()
            self.ex = None


RCCL_ERROR = None  # set when a weak-scaling run went on without the library communicator


def setup_rccl(ctx, dist: Dist, required: bool = True):
    """Library-level RCCL communicator (ds_comm_*) for the broadcast of the shared input and the
    gather of sharded results.  Returns False where RCCL is not asked for: one rank, a gloo
    rehearsal with several ranks on one GPU, or BENCH_BCAST=host (explicit opt-out: host
    broadcast + upload).  A communicator that HANGS in its set-up is a failure of the run: the
    diagnosis goes to stderr and the process exits non-zero.  One that reports an error is a failure
    too where the data path needs it (`required`: strong scaling gathers results over it); a weak-
    scaling run has no collective in its timed region, so it goes on with the host broadcast and
    says so in the JSON ("bcast": "host", "rccl_error": ...)."""
    global RCCL_ERROR
    if dist.world == 1:
        return False
    if dist.rehearsal:
        return False
    ident = C.create_string_buffer(128)
    ok = True
    if dist.rank == 0:
        ok = ctx.lib.ds_comm_unique_id(ident) == 0
    raw = dist.bcast_bytes(ident.raw, 128)
    if not dist.all_ok(ok):
        print(f"[bench] rank {dist.rank}: ds_comm_unique_id failed: {ctx.last_error()}", file=sys.stderr, flush=True)
        if required:
            sys.exit(3)
        RCCL_ERROR = f"ds_comm_unique_id failed: {ctx.last_error()}"
        return False
    # ncclCommInitRank is collective; a rank that never returns from it would hang the whole
    # run, so it runs under a watchdog -- which reports and exits non-zero, it does not continue
    import threading
    res = {}

    def init():
        res["rc"] = ctx.lib.ds_comm_init(ctx.handle, dist.world, dist.rank, raw)

    th = threading.Thread(target=init, daemon=True)
    th.start()
    limit = float(os.environ.get("BENCH_RCCL_TIMEOUT", "120"))
    th.join(timeout=limit)
    if th.is_alive():
        print(f"[bench] rank {dist.rank}/{dist.world}: ds_comm_init (ncclCommInitRank) has not returned after "
              f"{limit:.0f} s -- rendezvous over MASTER_ADDR={os.environ.get('MASTER_ADDR')} "
              f"NCCL_SOCKET_IFNAME={os.environ.get('NCCL_SOCKET_IFNAME')} "
              f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}; "
              "set NCCL_DEBUG=INFO for the transport log, or BENCH_BCAST=host to run without RCCL. "
              "Exiting with status 3.", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(3)
    ok = res.get("rc", -1) == 0
    if not ok:
        print(f"[bench] rank {dist.rank}: ds_comm_init failed: {ctx.last_error()}", file=sys.stderr, flush=True)
    if not dist.all_ok(ok):  # every rank takes the same way
        if required:
            sys.exit(3)
        RCCL_ERROR = "ds_comm_init failed on at least one rank" + ("" if ok else f": {ctx.last_error()}")
        return False
    return True


def shared_upload(ctx, dist: Dist, rccl: bool, arr: np.ndarray):
    """A device copy of `arr` on every rank, rank 0 owning the data: with the library's RCCL communicator rank 0
    uploads and ds_bcast sends it over xGMI (-> milliseconds of that collective); without it (one rank, ranks sharing a
    GPU, BENCH_BCAST=host) the host exchange carries the bytes and every rank uploads.  `arr` must have the same shape
    and dtype on every rank; only rank 0's values are used.  -> (DeviceBuffer, bcast_ms | None)"""
    from dsptoolbox_amd._lib import DeviceBuffer
    arr = np.ascontiguousarray(arr)
    buf = DeviceBuffer(ctx, max(arr.nbytes, 16))
    if rccl:
        if dist.rank == 0:
            ctx.upload(buf.ptr, arr)
        dist.barrier_sync(ctx)
        t0 = time.perf_counter()
        ctx.check(ctx.lib.ds_bcast(ctx.handle, C.c_void_p(buf.ptr), arr.nbytes, 0), "ds_bcast")
        ctx.sync()
        return buf, (time.perf_counter() - t0) * 1e3
    if dist.world > 1:
        from dsptoolbox_amd.distributed import broadcast_array
        arr = broadcast_array(arr if dist.rank == 0 else None, src=0)
    ctx.upload(buf.ptr, arr)
    return buf, None


def shard_of(n_units: int, shard):
    from dsptoolbox_amd.distributed import shard_range
    if shard is None:
        return 0, n_units
    return shard_range(n_units, shard[1], shard[0])


def gather_result(ctx, rccl: bool, shard, d_local, d_all, bytes_per_rank: int):
    """Strong scaling: all-gather the result slices of one step (padded to the largest shard)."""
    if rccl and shard is not None and shard[1] > 1:
        ctx.check(ctx.lib.ds_allgather(ctx.handle, C.c_void_p(d_local.ptr), C.c_void_p(d_all.ptr),
                                       bytes_per_rank), "ds_allgather")


# ---------------------------------------------------------------------------
# Every workload maker builds the step of ONE rank: `shard` = None (the whole job, also each
# rank's independent batch under weak scaling) or (rank, world) for that rank's part of the one
# job under strong scaling.  Returns (step, units, algorithmic bytes or flops, bound, info,
# cpu_baselines, bcast_ms, dominant kernel names); units / bytes are those of the WHOLE job.
def welch_h1(args, ctx, dist, shard, rccl, W=4096):
    from dsptoolbox_amd import backend
    from dsptoolbox_amd._lib import DeviceBuffer
    from dsptoolbox_amd.generators import sweep_and_responses
    from dsptoolbox_amd.standard.enums import SpectrumScaling, Window

    n, n_cy = 2**20, 64
    x, y = sweep_and_responses(n, n_cy, FS)
    if shard is None and dist.rank > 0:  # weak scaling: another batch per rank
        rng = np.random.default_rng(9000 + dist.rank)
        y = y[:, rng.permutation(n_cy)] * (1.0 + 0.01 * dist.rank)
    a, b = shard_of(n_cy, shard)
    n_loc = b - a
    n_max = -(-n_cy // (shard[1] if shard else 1))
    window = backend._window_array(Window.Hann, W)
    hop, n_frames = backend._welch_framing(n, W, 50, window)
    amp, norm_scale, factor, phys = backend._finish_params(SpectrumScaling.FFTBackward, W, FS, window)
    d_y = DeviceBuffer.from_array(ctx, backend._planar_f32(y[:, a:b])) if n_loc else None
    d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
    d_x, bcast_ms = shared_upload(ctx, dist, rccl, backend._planar_f32(x))  # the shared sweep channel
    B = W // 2 + 1
    # result slice of this rank: tf (B, n_loc) complex64 then coh (B, n_loc) float32, one buffer
    slot = B * n_max * 12
    d_res = DeviceBuffer(ctx, max(slot, 16))
    d_all = DeviceBuffer(ctx, slot * shard[1]) if (shard and shard[1] > 1) else None

    def step():
        if n_loc:
            ctx.check(ctx.lib.ds_welch_tf_dev(
                ctx.handle, C.c_void_p(d_x.ptr), 1, n, C.c_void_p(d_y.ptr), n_loc, n, n, W, hop, n_frames,
                C.c_void_p(d_w.ptr), int(args.detrend), 0, 1, amp, norm_scale, factor, phys,
                C.c_void_p(d_res.ptr), C.c_void_p(d_res.ptr + B * n_loc * 8)), "ds_welch_tf_dev")
        gather_result(ctx, rccl, shard, d_res, d_all, slot)

    samples_per_step = (n_cy + 1) * n
    alg_bytes = (n_cy + 1) * n * 4 + B * n_cy * 8 + B * n_cy * 4
    info = dict(
        workload=f"welch_h1: 64 output ch + 1 sweep input ch x 2^20 samples, nfft {W}, Hann, 50% overlap"
                 + (", detrend" if args.detrend else ""),
        channels=n_cy, samples_per_channel=n, nfft=W, overlap_percent=50, frames=n_frames)

    def verify():
        raw = d_res.to_array((B * n_loc * 12,), np.uint8)
        tf = raw[:B * n_loc * 8].view(np.complex64).reshape(B, n_loc)
        coh = raw[B * n_loc * 8:].view(np.float32).reshape(B, n_loc)
        assert np.all(np.isfinite(tf[1:]))
        return tf, coh

    def cpu_reference_loop():
        from oracle import dsp_oracle as orc
        cc = min(args.cpu_channels, n_cy)
        reps = 2  # ~14 s of single-core work for the full 64 channels
        if args.cpu_bounded:  # an entry of "workloads": a quarter of the channels, once
            cc, reps = min(cc, 16), 1
        t0 = time.perf_counter()
        for _ in range(reps):
            rt, rc = orc.compute_transfer_function(y[:, :cc], x, FS, W, "H1", detrend=bool(args.detrend))
        dt = (time.perf_counter() - t0) / reps
        tf, coh = verify()
        fr = np.fft.rfftfreq(W, 1 / FS)
        sl = (fr >= 30.0) & (fr <= 19000.0)  # bins the 20 Hz - 20 kHz sweep excites
        err = max(orc.rel_max(tf[sl, :cc], rt[sl]), orc.rel_max(coh[sl, :cc], rc[sl]))
        # the reference loop re-frames/re-FFTs x per output channel: samples consumed = (cc + 1) n
        return dict(value=(cc + 1) * n / dt / 1e6, unit="Msamples/s", cores=1, kind="port",
                    sample=f"oracle.compute_transfer_function (reference loop structure) on {cc} of "
                           f"{n_cy} output channels x 2^20, {reps} passes of {dt:.1f} s",
                    parity_rel_max_vs_gpu=err)

    def cpu_batched():
        from oracle import dsp_oracle as orc
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            rt, rc = orc.compute_transfer_function_batched(y, x, FS, W, "H1", detrend=bool(args.detrend),
                                                           workers=-1)
        dt = (time.perf_counter() - t0) / reps
        return dict(value=(n_cy + 1) * n / dt / 1e6, unit="Msamples/s", cores=os.cpu_count(), kind="port",
                    sample=f"oracle.compute_transfer_function_batched (one framing + one rFFT batch per signal, "
                           f"scipy.fft workers=-1), all {n_cy} channels, {reps} passes of {dt:.1f} s")

    return step, samples_per_step, alg_bytes, "hbm", info, (cpu_reference_loop, cpu_batched), bcast_ms, \
        ("welch4096_main", "welch1024_main", "welch_yacc")


def fir_bank(args, ctx, dist, shard, rccl):
    from dsptoolbox_amd import backend
    from dsptoolbox_amd._lib import DeviceBuffer
    from dsptoolbox_amd.generators import fir_bank_taps

    n, n_ch, K, T = 2**22, 8, 32, 4097
    seed = 3 + (dist.rank if shard is None else 0)
    x = np.random.default_rng(seed).standard_normal((n, n_ch)) * 0.1
    taps = fir_bank_taps(K, T, FS).astype(np.float32)
    a, b = shard_of(K, shard)  # Parallel mode shards by bands
    k_loc = b - a
    d_x = DeviceBuffer.from_array(ctx, backend._planar_f32(x))
    # the shared taps: rank 0 owns the bank, every rank receives all of it (RCCL broadcast over xGMI when the
    # library's communicator is up) and filters with its own bands [a, b)
    d_taps, bcast_ms = shared_upload(ctx, dist, rccl, taps)
    d_y = DeviceBuffer(ctx, max(k_loc, 1) * n_ch * n * 4)

    def step():  # the 4 GiB result stays sharded (band-major): no gather in the reference either
        if k_loc:
            ctx.check(ctx.lib.ds_fir_ola_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n,
                                             C.c_void_p(d_taps.ptr + 4 * a * T), k_loc, T, backend.DS_FB_PARALLEL,
                                             C.c_void_p(d_y.ptr), n), "ds_fir_ola_dev")

    alg_bytes = n_ch * n * 4 + K * T * 4 + K * n_ch * n * 4
    info = dict(workload="fir_bank: FilterBank Parallel, 32 x 4097-tap FIR over 8 ch x 2^22",
                channels=n_ch, samples_per_channel=n, bands=K, taps=T)

    def cpu_baseline():
        from oracle import dsp_oracle as orc
        err = 0.0
        bands = list(range(a, b, 4 if args.cpu_bounded else 1))  # an entry of "workloads": every 4th band
        t0 = time.perf_counter()
        for k in bands:  # the reference loop: one oaconvolve per band (~0.5 s each)
            ref = orc.lfilter_fir(taps[k].astype(np.float64), x)
            if True:  # parity on every band the CPU leg computes (the download is outside what is measured)
                t1 = time.perf_counter()
                got = np.empty((n_ch, n), dtype=np.float32)
                ctx.download(d_y.ptr + 4 * (k - a) * n_ch * n, got)
                err = max(err, orc.rel_max(got.T, ref))
                t0 += time.perf_counter() - t1
        dt = (time.perf_counter() - t0) * (b - a) / max(1, len(bands))  # scaled to the whole bank
        return dict(value=n_ch * n / dt / 1e6, unit="Msamples/s", cores=1, kind="port",
                    sample=f"oracle.lfilter_fir (scipy oaconvolve), {len(bands)} of {K} bands timed "
                           f"({dt * len(bands) / max(1, b - a):.1f} s), scaled to all {K}; parity on each of them",
                    parity_rel_max_vs_gpu=err)

    return step, n_ch * n, alg_bytes, "hbm", info, (cpu_baseline,), bcast_ms, ("fir",)


def csm(args, ctx, dist, shard, rccl):
    from dsptoolbox_amd import backend
    from dsptoolbox_amd._lib import DeviceBuffer
    from dsptoolbox_amd.generators import mic_array_noise
    from dsptoolbox_amd.standard.enums import SpectrumScaling, Window

    n, n_ch, W = 512000, 64, 1024
    x = mic_array_noise(n, n_ch, 4 + (dist.rank if shard is None else 0))
    window = backend._window_array(Window.Hann, W)
    hop, n_frames = backend._welch_framing(n, W, 50, window)
    amp, norm_scale, factor, phys = backend._finish_params(SpectrumScaling.FFTBackward, W, FS, window)
    d_x = DeviceBuffer.from_array(ctx, backend._planar_f32(x))
    d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
    B = W // 2 + 1
    a, b = shard_of(B, shard)  # the matrix shards by frequency bins (all channel pairs are needed)
    b_max = -(-B // (shard[1] if shard else 1))
    slot = b_max * n_ch * n_ch * 8
    d_c = DeviceBuffer(ctx, slot)
    d_all = DeviceBuffer(ctx, slot * shard[1]) if (shard and shard[1] > 1) else None

    def step():
        if shard is None:
            ctx.check(ctx.lib.ds_csm_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n, W, hop, n_frames,
                                         C.c_void_p(d_w.ptr), 1, 0, amp, norm_scale, factor, phys,
                                         C.c_void_p(d_c.ptr)), "ds_csm_dev")
        elif b > a:
            ctx.check(ctx.lib.ds_csm_bins_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n, W, hop, n_frames,
                                              C.c_void_p(d_w.ptr), 1, amp, norm_scale, factor, phys, a, b - a,
                                              C.c_void_p(d_c.ptr)), "ds_csm_bins_dev")
        gather_result(ctx, rccl, shard, d_c, d_all, slot)

    flops = B * n_ch * n_ch * n_frames * 8.0
    # Two HBM-streaming kernels since the Gram product runs on the bf16 matrix pipe (three bf16
    # pieces per fp32 value, kernels_csm_b3.hpp): the transform reads the samples once and writes the
    # spectrogram X[bin][frame][mic]; the product reads X once and writes the matrices.  "step" is
    # what SURVEY 8(d) counts for the fused ideal (samples in, matrices out).
    x_bytes = B * n_frames * n_ch * 8.0
    alg = {"stft": n_ch * n * 4.0 + x_bytes, "csm_gemm": x_bytes + B * n_ch * n_ch * 8.0,
           "step": n_ch * n * 4.0 + B * n_ch * n_ch * 8.0, "gemm_flops": flops,
           # 64 000 framed rFFTs of 1024 points at 2.5 W log2 W
           "fft_flops": n_ch * n_frames * 2.5 * W * np.log2(W)}
    info = dict(workload="csm: 64-mic Welch cross-spectral matrix, nfft 1024, 1000 frames",
                channels=n_ch, samples_per_channel=n, nfft=W, frames=n_frames,
                gemm_flops="513*64*64*1000*8 (full Hermitian count, fp32-equivalent)")

    def cpu_baseline():
        from oracle import dsp_oracle as orc
        reps = 1 if args.cpu_bounded else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            f, ref = orc.csm_welch_batched(x, FS, W, "hann", 50, True, "FFTBackward", workers=-1)
        dt = (time.perf_counter() - t0) / reps
        got = d_c.to_array((b - a, n_ch, n_ch), np.complex64)
        lo = max(a, 1)
        return dict(value=n_ch * n / dt / 1e6, unit="Msamples/s", cores=os.cpu_count(), kind="port",
                    sample=f"oracle.csm_welch_batched (batched restatement, scipy.fft workers=-1), all 64 "
                           f"mics, {reps} passes of {dt:.1f} s; the reference's 2080-pair loop takes ~100 s",
                    parity_rel_max_vs_gpu=orc.rel_max(got[lo - a:], ref[lo:b]))

    return step, n_ch * n, alg, "hbm", info, (cpu_baseline,), None, ("stft", "csm_gemm")


def deconv(args, ctx, dist, shard, rccl):
    from dsptoolbox_amd import backend
    from dsptoolbox_amd._lib import DeviceBuffer
    from dsptoolbox_amd.generators import exponential_sweep
    from dsptoolbox_amd.transfer_functions import (_inverse_hann_band, find_frequencies_above_threshold,
                                                   find_nearest_points_index_in_vector)

    n, items_all, n_ch = 8192, 1024, 2
    B = n // 2 + 1
    # the shared reference sweep and 1024 stereo responses to it: y[i, c] = x (*) h_ic + noise, h a 256-tap decaying
    # Gaussian-noise response (circular convolution: the reference divides whole-signal spectra)
    x = exponential_sweep(n, FS)
    rng = np.random.default_rng(5000 + (dist.rank if shard is None else 0))
    h = rng.standard_normal((items_all, n_ch, 256)) * np.exp(-np.arange(256) / 40.0)
    y = np.fft.irfft(np.fft.rfft(x)[None, None, :] * np.fft.rfft(h, n, axis=-1), n, axis=-1)
    y = (y + 1e-3 * rng.standard_normal(y.shape)).astype(np.float32)
    del h
    a, b = shard_of(items_all, shard)  # independent items shard
    items = b - a
    y = y[a:b]
    d_y = DeviceBuffer.from_array(ctx, y) if items else None
    # The regularised inverse of the sweep, conj(X) / (|X|^2 + eps) with the reference's band detection
    # (transfer_functions.py:152-167, _transfer_functions.py:31-35), built ONCE by rank 0 through the package's own
    # device path (ds_rfft -> ds_deconv_inverse) and handed to every rank: RCCL broadcast over xGMI (BASELINE config 5)
    r = np.zeros(B, dtype=np.complex64)
    if dist.rank == 0:
        den = backend.rfft_spectrum(x[:, None], n)
        freqs = np.fft.rfftfreq(n, 1 / FS)
        lo_hi = find_frequencies_above_threshold(den[:, 0], freqs, -30.0)
        band = np.array([lo_hi[0] / np.sqrt(2), lo_hi[0], lo_hi[1], np.min([lo_hi[1] * np.sqrt(2), FS / 2])])
        eps = _inverse_hann_band(find_nearest_points_index_in_vector(band, freqs), B) * 10 ** (30 / 20)
        r = np.ascontiguousarray(backend.regularized_inverse(den, eps)[:, 0].astype(np.complex64))
    d_r, bcast_ms = shared_upload(ctx, dist, rccl, r)
    d_o = DeviceBuffer(ctx, max(y.nbytes, 16))

    def step():  # the impulse responses stay with the rank that owns the items
        if items:
            ctx.check(ctx.lib.ds_deconv_dev(ctx.handle, C.c_void_p(d_y.ptr), items, n_ch, n, n, n,
                                            C.c_void_p(d_r.ptr), 0, n, n, C.c_void_p(d_o.ptr)),
                      "ds_deconv_dev")

    alg_bytes = 2 * items_all * n_ch * n * 4 + r.nbytes
    info = dict(workload="deconv: stereo spectral deconvolutions n=8192 against a shared regularised inverse sweep",
                items=items_all, channels=n_ch, samples_per_channel=n)

    def cpu_baseline():
        # the reference's own call, one per stereo item (transfer_functions.py:61-184: spectrum of the sweep, band
        # detection, regularised division, irfft) on a bounded sample of items spread over the batch; every one of
        # them is compared with the device result
        from oracle import dsp_oracle as orc
        count = min(items, 32 if args.cpu_bounded else 256)
        pick = np.unique(np.linspace(0, items - 1, count).astype(int))
        got_all = d_o.to_array((items, n_ch, n), np.float32)
        xx = x[:, None]
        err, refs = 0.0, []
        t0 = time.perf_counter()
        for i in pick:
            refs.append(orc.spectral_deconvolve(y[i].T.astype(np.float64), xx, FS))
        dt = time.perf_counter() - t0
        for i, ref in zip(pick, refs):
            err = max(err, orc.rel_max(got_all[i].T, ref))
        return dict(value=len(pick) * n_ch * n / dt / 1e6, unit="Msamples/s", cores=1, kind="port",
                    sample=f"oracle.spectral_deconvolve per stereo item (reference call structure: sweep spectrum, band "
                           f"detection, regularised division, irfft), {len(pick)} of {items} items spread over the batch "
                           f"in {dt:.2f} s; parity on every one of them",
                    parity_rel_max_vs_gpu=err)

    return step, items_all * n_ch * n, alg_bytes, "hbm", info, (cpu_baseline,), bcast_ms, ("deconv",)


def pmc_summary(workload: str, kernel_hints):
    """Counters of the dominant kernel from the committed rocprofv3 PMC summary of this same
    command (profiles/rNN_<workload>_rocprofv3_summary.txt, latest round) -> ({counter: per-launch
    average}, file name) or ({}, None)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_rocprofv3_summary.txt")))
    if not files:
        return {}, None
    text = open(files[-1]).read()
    sect = text.split("== PMC", 1)[-1].split("== bench line", 1)[0]
    blocks = re.split(r"\n(?=\S)", sect)
    for hint in kernel_hints:
        for blk in blocks:
            head = blk.splitlines()[0] if blk.strip() else ""
            if hint not in head:
                continue
            vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\w+)\s+([0-9.]+)", blk, re.M)}
            if vals:
                return vals, os.path.basename(files[-1])
    return {}, None


def pmc_kernel_current(workload: str, kernel_hints):
    """Were the committed counters taken on the machine code that runs now?  The summary file records a fingerprint
    of each profiled kernel (tools/prof_summary.py, "== kernel code"); this compares the dominant kernel's with the
    library in use.  True / False, or None when either side has no fingerprint (an older file, no c++filt)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_rocprofv3_summary.txt")))
    if not files:
        return None
    text = open(files[-1]).read()
    if "== kernel code (" not in text:
        return None
    sect = text.split("== kernel code (", 1)[1].split("\n==", 1)[0]
    recorded = {m.group(2).strip(): m.group(1) for m in re.finditer(r"^\s+([0-9a-f]{16})\s+(\S.*)$", sect, re.M)}
    from dsptoolbox_amd import _build
    m = re.search(r"^== compiler of the profiled library: (.*)$", text, re.M)
    if m and m.group(1).strip() != _build.compiler_id():
        return None  # a library built by another compiler: the fingerprints say nothing either way
    now = _build.demangled_fingerprints()
    if not now:
        return None
    for hint in kernel_hints:
        for name, h in recorded.items():
            if hint in name:
                cur = [v for d, v in now.items() if d[:60] == name]
                return len(cur) == 1 and cur[0] == h
    return None


# ---------------------------------------------------------------------------
EVENT_STRIDE = 4  # the dominant kernel is bracketed at every 4th launch of the timed region


def timed_steps(ctx, dist, step, steps: int, events: bool, dominant):
    """W. warm-up done by the caller.  Times exactly `steps` steps between two
    barrier + synchronize brackets; only the dominant kernel is event-bracketed inside, at every
    EVENT_STRIDE-th launch (timestamped dispatches cost stream time: stamping every launch adds a few % to
    the step it measures)."""
    dist.barrier_sync(ctx)
    ctx.profile_only(dominant)  # None (no instrumented warm-up step): every kernel
    ctx.profile_stride(EVENT_STRIDE if dominant else 1)
    ctx.profile_enable(events)
    ctx.profile_report()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(steps):
        step()
    ev_ms = ctx.timer_stop()
    dist.barrier_sync(ctx)
    wall = time.perf_counter() - t0
    prof = ctx.profile_report()
    ctx.profile_enable(False)
    ctx.profile_only(None)
    ctx.profile_stride(1)
    return wall, ev_ms, prof


def preheat_steps(ctx, step, ms: float):
    """Run `step` back to back, untimed, for about `ms` milliseconds of wall time -> {"steps", "ms"}."""
    n, t0 = 0, time.perf_counter()
    while ms > 0 and (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(20):
            step()
        ctx.sync()
        n += 20
    return dict(steps=n, ms=(time.perf_counter() - t0) * 1e3)


def launch_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` with no launcher environment: start the N ranks here, as children of a
    process that has not touched the GPU -- one rank per GPU, the launcher's usual environment (RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT) --, pass their output through
    and return the first non-zero exit status (the other ranks are stopped then)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    print(f"[bench] no launcher environment: starting {n_gpus} ranks of {os.path.basename(__file__)} "
          f"(rendezvous 127.0.0.1:{port})", file=sys.stderr, flush=True)
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:  # a rank failed: the others would wait for it in the next barrier
                    q.terminate()
    return rc


def measure(args, ctx, dist, workload: str, steps: int, warmup: int, strong: bool, rccl: bool):
    """Time one workload on this rank's GPU: warm-up (its last step bracketed kernel by kernel), then exactly
    `steps` steps between two barrier + synchronize brackets with the dominant kernel's own dispatch
    timestamps sampled inside.  -> (result dict on rank 0 | None, maker tuple, wall seconds)."""
    shard = (dist.rank, dist.world) if strong else None
    maker = dict(welch_h1=welch_h1, welch_h1_1024=lambda a, c, d, sh, r: welch_h1(a, c, d, sh, r, W=1024),
                 fir_bank=fir_bank, csm=csm, deconv=deconv)[workload]
    made = maker(args, ctx, dist, shard, rccl)
    step, units, alg, bound, info, cpu_legs, bcast_ms, dominant = made
    info["parallelism"] = (f"strong: one job sharded x{dist.world}" if strong
                           else f"weak: independent batch x{dist.world}")
    # Warm-up; its last step is bracketed kernel by kernel (HIP events on the library's stream) to
    # find the dominant kernel and the per-kernel breakdown.  An event pair costs ~3 us of stream
    # time, so in the timed region only the dominant kernel is bracketed.
    events = not os.environ.get("BENCH_NO_KERNEL_EVENTS")
    prof_all = {}
    # Clock ramp: an MI355X that has been idle (the set-up above is host work) starts its first milliseconds of
    # kernels at a low engine clock -- the same binary reads 105-107 us per launch in its first 8 ms and 91-93 us from
    # then on (profiles/r05_clock_ramp.txt).  W = 5 warm-up steps are 0.6 ms.  So the step runs untimed for
    # --preheat-ms first (reported as "preheat"); the W warm-up steps and the K timed steps follow unchanged.
    preheat = preheat_steps(ctx, step, args.preheat_ms)
    for i in range(warmup):
        if events and i == warmup - 1:
            ctx.sync()
            ctx.profile_enable(True)
            ctx.profile_report()
            step()
            prof_all = ctx.profile_report()
            ctx.profile_enable(False)
        else:
            step()
    dom = max((k for k in dominant if k in prof_all), key=lambda k: prof_all[k][0], default=None)
    if dom is None and prof_all:
        dom = max(prof_all, key=lambda k: prof_all[k][0])
    wall, ev_ms, prof = timed_steps(ctx, dist, step, steps, events, dom)
    wall = dist.max_over_ranks(wall)
    # a second, longer region right behind the one the command line asks for: K = 20 steps are
    # 2.5 ms and 5 event brackets; this one is --steady-steps steps and a quarter as many brackets
    steady = None
    if args.steady_steps > 0 and getattr(args, "steady_now", True):
        s_wall, s_ev, s_prof = timed_steps(ctx, dist, step, args.steady_steps, events, dom)
        s_wall = dist.max_over_ranks(s_wall)
        steady = dict(steps=args.steady_steps, ms_per_step=s_wall * 1e3 / args.steady_steps,
                      step_event_ms=s_ev / args.steady_steps,
                      value=units * (1 if strong else dist.world) / (s_wall / args.steady_steps) / 1e6, unit="Msamples/s")
        if s_prof and dom in s_prof:
            steady["kernel"] = dom
            steady["kernel_avg_ms"] = s_prof[dom][0] / s_prof[dom][1]
            steady["kernel_brackets"] = s_prof[dom][1]
    if dist.rank != 0:
        return None, made, wall
    ms_per_step = wall * 1e3 / steps
    # weak: every rank ran the whole job's units; strong: the ranks shared them
    value = units * (1 if strong else dist.world) / (wall / steps) / 1e6
    if not prof:  # BENCH_NO_KERNEL_EVENTS=1: step time without the per-kernel event markers (dev)
        return {"value": value, "ms_per_step": ms_per_step, "step_event_ms": ev_ms / steps,
                "note": "no per-kernel events: no roofline"}, made, wall
    if dom is None or dom not in prof:
        dom = next((k for k in dominant if k in prof), max(prof, key=lambda k: prof[k][0]))
    # roofline.frac / achieved / kernel_avg_ms: the dominant kernel's own begin / end timestamps (the two
    # events ride on its dispatch packet, hipExtLaunchKernel -- what rocprofv3's kernel trace reports),
    # averaged over the sampled launches of the timed region.  Nothing is subtracted.
    dom_ms = prof[dom][0] / prof[dom][1]
    launches_per_step = max(1, round(prof[dom][1] * EVENT_STRIDE / steps))
    # algorithmic work of ONE launch on THIS rank (strong scaling: its share of the job)
    alg_parts = alg if isinstance(alg, dict) else None  # per kernel (csm: two streaming kernels)
    if alg_parts:
        alg = alg_parts["step"]
    div = dist.world if strong else 1
    alg_launch = (alg_parts[dom] if alg_parts else alg) / launches_per_step / div
    if alg_parts and dom == "stft":  # the bin shards of a strong-scaling run all transform every frame
        alg_launch = alg_parts[dom] / launches_per_step
    step_ms = ev_ms / steps
    achieved = alg_launch / (dom_ms * 1e-3) / 1e9
    hbm = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=None)
    # the second denominator SURVEY 8(d) asks for: what a plain copy reaches on THIS GPU
    gbs = C.c_double(0.0)
    if ctx.lib.ds_measure_copy(ctx.handle, 1 << 30, 10, C.byref(gbs)) == 0 and gbs.value > 0:
        hbm["measured_copy_gbs"] = gbs.value
        hbm["frac_of_measured_copy"] = achieved / gbs.value
    if alg_parts:
        # config 4, SURVEY 8(d): algorithmic flops (framed rFFTs + the full Hermitian count of the Gram products) over
        # the STEP against the fp32 matrix peak; the two kernels' HBM fractions follow under "kernels"
        flops = (alg_parts["gemm_flops"] + alg_parts["fft_flops"]) / div
        tf = flops / (step_ms * 1e-3) / 1e12
        step_gbs = alg_parts["step"] / div / (step_ms * 1e-3) / 1e9
        roof = dict(bound="mfma", achieved=tf, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / MFMA_F32_PEAK_TFLOPS,
                    traffic=None, algorithmic_flops_per_step=flops, time_base="step (both kernels), HIP events",
                    # the same step by BYTES (samples in, matrices out: SURVEY 8(d)'s fused ideal) against the HBM roofline
                    step_bytes_gbs=step_gbs, step_bytes_frac=step_gbs / HBM_PEAK_GBS,
                    dominant_kernel_hbm=hbm)
    else:
        roof = hbm
    hints = KERNEL_HINTS.get(dom, (dom,))
    pmc, src = pmc_summary(workload, hints)
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # KiB, separate --pmc passes; on gfx950 FETCH_SIZE counts half of a coalesced streaming
        # read (MI355X_MICROARCH.md, HBM section), hence the factor 2
        roof["traffic"] = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        roof["traffic_source"] = src + " (2*FETCH_SIZE + WRITE_SIZE) KiB"
        current = pmc_kernel_current(workload, hints)
        if alg_parts:  # both kernels of the step against the step's algorithmic bytes
            tot = 0.0
            for h in (("k_stft",), ("k_csm_gemm",)):
                kp, _ = pmc_summary(workload, h)
                tot += (2.0 * kp.get("FETCH_SIZE", 0.0) + kp.get("WRITE_SIZE", 0.0)) * 1024.0
            roof["traffic"] = tot
            roof["traffic_source"] = src + " (2*FETCH_SIZE + WRITE_SIZE) KiB, transform + Gram kernel"
            both = [pmc_kernel_current(workload, h) for h in (("k_stft",), ("k_csm_gemm",))]
            current = None if None in both else all(both)
        # the counters are a committed file, not this run: the file carries a fingerprint of the
        # machine code they were taken on; a kernel rebuilt since then gets no traffic figure
        roof["traffic_kernel_current"] = current
        if current is False:
            roof["traffic_stale"] = dict(traffic=roof["traffic"], note="counters of an older build of this kernel: re-run tools/prof_all.sh")
            roof["traffic"] = None
    if "SQ_INSTS_VALU" in pmc:
        # the ceiling the vector pipe puts on this kernel: wave-level VALU instructions of one
        # launch x 2 cycles (wave64 on a SIMD-32) over 1024 SIMDs; at the 2.4 GHz maximum clock --
        # under load the chip holds 1.8-2.0 GHz (s_memtime / s_memrealtime in the kernel)
        floor_ms = pmc["SQ_INSTS_VALU"] * 2.0 / 1024.0 / 2.4e9 * 1e3
        roof["valu_issue"] = dict(insts_per_launch=pmc["SQ_INSTS_VALU"], floor_ms_at_2p4_ghz=floor_ms,
                                  valu_issue_frac=floor_ms / dom_ms, source=src)
    if workload == "welch_h1" and dom == "welch4096_main":
        # the demonstrated ceiling of this instruction stream: the kernel with every load, LDS access and
        # barrier removed (VALU only) on the same chip, profiles/r04_welch_ceiling.txt
        roof["ceiling_ms"] = WELCH_CEILING_MS
        roof["ceiling_source"] = "profiles/r04_welch_ceiling.txt (tools/exp/exp_w4.hip -DW4_AB=15, steady state)"
        roof["ceiling_frac"] = alg_launch / (WELCH_CEILING_MS * 1e-3) / 1e9 / HBM_PEAK_GBS
    if alg_parts and prof_all:
        # every kernel of the step against the HBM roofline (warm-up step, every kernel bracketed), and
        # the Gram product's matrix-pipe numbers: fp32-equivalent flops by the full Hermitian count, and
        # the bf16 instructions really issued (v_mfma_f32_32x32x16_bf16 = 32768 flop; 60 per 16 frames
        # and workgroup: 10 of the 16 tile products, 6 piece products each)
        roof["kernels"] = {}
        for k in ("stft", "csm_gemm"):
            if k in prof_all:
                nbytes = alg_parts[k] / (1 if k == "stft" else div)
                kg = nbytes / (prof_all[k][0] * 1e-3) / 1e9
                roof["kernels"][k] = dict(ms=prof_all[k][0], algorithmic_bytes=nbytes, gbs=kg, frac=kg / HBM_PEAK_GBS)
        if "csm_gemm" in prof_all:
            g_ms = prof_all["csm_gemm"][0]
            gpmc, gsrc = pmc_summary(workload, ("k_csm_gemm",))
            gemm = dict(ms=g_ms, fp32_equivalent_tflops=alg_parts["gemm_flops"] / div / (g_ms * 1e-3) / 1e12,
                        fp32_mfma_peak_tflops=MFMA_F32_PEAK_TFLOPS,
                        note="fp32-exact product from three bf16 pieces per value on the bf16 matrix pipe; "
                             "bound by the operand stream from HBM, not by the pipe")
            if gpmc.get("SQ_INSTS_MFMA"):
                ex = gpmc["SQ_INSTS_MFMA"] * 32768.0
                gemm.update(mfma_insts_per_launch=gpmc["SQ_INSTS_MFMA"], executed_bf16_tflops=ex / (g_ms * 1e-3) / 1e12,
                            frac_of_bf16_peak=ex / (g_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, source=gsrc)
            roof["gemm"] = gemm
    roof["kernel"] = dom
    roof["kernel_event_sampling"] = f"every {EVENT_STRIDE}th launch of the timed region ({prof[dom][1]} brackets)"
    roof["kernel_avg_ms"] = dom_ms
    roof["kernel_time_source"] = ("begin / end timestamps of the kernel's own dispatch (HIP events passed to hipExtLaunchKernel on the "
                                  "library's stream); the rocprofv3 kernel-trace average of the same command is in profiles/")
    roof["algorithmic_per_launch"] = alg_launch
    out = {
        "metric": "Msamples/s + GB/s vs HBM roofline, 64ch Welch H1 nfft=4096 @1/2/4/8 GPU"
                  if workload == "welch_h1" else f"Msamples/s ({workload})",
        "value": value, "unit": "Msamples/s", "n_gpus": dist.world, "steps": steps,
        "warmup": warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": info, "roofline": roof,
        "step_event_ms": step_ms,
        "kernels_ms_per_step": ({k: v[0] for k, v in prof_all.items()} if prof_all
                                else {k: v[0] / steps for k, v in prof.items()}),
        "kernels_ms_per_step_source": ("last warm-up step, every kernel bracketed" if prof_all
                                       else "timed region"),
        "whole_step_gbs": alg / div / (step_ms * 1e-3) / 1e9,
        "algorithmic_bytes_per_step": alg / div,
    }
    out["preheat"] = dict(preheat, note="untimed steps in front of the W warm-up steps: the engine clock of an idle "
                                        "chip ramps for ~10 ms (profiles/r05_clock_ramp.txt)")
    if steady:
        if roof["bound"] == "mfma":  # the same fraction over the longer region: flops of the step / its time / peak
            steady["roofline_frac"] = roof["frac"] * step_ms / steady["step_event_ms"]
        elif "kernel_avg_ms" in steady:
            steady["roofline_frac"] = roof["frac"] * dom_ms / steady["kernel_avg_ms"]
        out["steady_state"] = steady
    if bcast_ms is not None:
        out["rccl_bcast_ms"] = bcast_ms
    return out, made, wall


WELCH_CEILING_MS = 0.0645  # VALU-only build of welch4096::k_y3, profiles/r04_welch_ceiling.txt


def workload_entry(args, ctx, dist, name: str):
    """One entry of the default line's "workloads": the same measurement over a shorter timed region, the
    fraction by SURVEY 8(d)'s own definition for that config, and a BOUNDED CPU leg with its parity."""
    import gc
    args.cpu_bounded = True
    args.steady_now = False  # these entries already time --workload-steps (200) steps
    out, made, _ = measure(args, ctx, dist, name, args.workload_steps, 20, False, False)
    args.steady_now = True
    roof = out["roofline"]
    alg = roof.get("algorithmic_flops_per_step") if roof["bound"] == "mfma" else roof["algorithmic_per_launch"]
    ent = dict(workload=out["config"]["workload"], ms_per_step=out["ms_per_step"], step_event_ms=out["step_event_ms"],
               value=out["value"], unit=out["unit"], steps=args.workload_steps,
               kernel=roof["kernel"], kernel_avg_ms=roof["kernel_avg_ms"], kernels_ms_per_step=out["kernels_ms_per_step"],
               bound=roof["bound"], frac=roof["frac"], achieved=roof["achieved"], peak=roof["peak"], roofline_unit=roof["unit"],
               frac_definition=("algorithmic flops (framed rFFTs + full Hermitian Gram count) / step time / 157.3 TFLOP/s fp32 matrix peak"
                                if roof["bound"] == "mfma" else
                                "algorithmic bytes of one launch / dominant kernel's dispatch time / 8 TB/s"))
    if roof.get("traffic"):
        base = out["algorithmic_bytes_per_step"] if roof["bound"] == "mfma" else roof["algorithmic_per_launch"]
        ent["traffic_ratio"] = roof["traffic"] / base
        ent["traffic_source"] = roof.get("traffic_source")
    else:
        ent["traffic_ratio"] = None
    ent["traffic_kernel_current"] = roof.get("traffic_kernel_current")
    if roof["bound"] == "mfma":
        ent["step_bytes_frac"] = roof.get("step_bytes_frac")
        ent["kernels_hbm_frac"] = {k: v["frac"] for k, v in roof.get("kernels", {}).items()}
        if "gemm" in roof and "frac_of_bf16_peak" in roof["gemm"]:
            ent["executed_bf16_frac"] = roof["gemm"]["frac_of_bf16_peak"]
            ent["executed_bf16_tflops"] = roof["gemm"]["executed_bf16_tflops"]
    cpu = made[5][0]()
    cpu["host_cpus"] = os.cpu_count()
    ent["parity_rel_max_vs_oracle"] = cpu.pop("parity_rel_max_vs_gpu", None)
    ent["cpu_baseline"] = cpu
    args.cpu_bounded = False
    del made, out
    gc.collect()  # the workload's device buffers go with its closures
    return ent


def short_estimate_entry():
    """The precision route of a SHORT estimate as a number: the reference's default
    window (1024 samples) on one second of 64-channel audio at 48 kHz is 94 frames -- fewer than 128 --, so the
    reference-shaped API (backend.welch_transfer_function, host float64 arrays in and out) sends it through the float64
    kernels (ds_welch_tf_x64); the same call forced onto the fp32 kernels beside it.  PCIe-inclusive wall time, median
    of 7 calls each after 2 warm-up calls; both results against the oracle."""
    from dsptoolbox_amd import backend
    from dsptoolbox_amd.generators import sweep_and_responses
    from oracle import dsp_oracle as orc
    n, n_cy, W = 48000, 64, 1024
    x, y = sweep_and_responses(n, n_cy, FS)
    rt, rc = orc.compute_transfer_function(y, x, FS, W, "H1")
    fr = np.fft.rfftfreq(W, 1 / FS)
    band = (fr >= 30.0) & (fr <= 19000.0)
    ent = dict(workload=f"welch_h1 through the API, {n_cy} + 1 channels x {n} samples, window {W}: "
                        f"{backend._welch_framing(n, W, 50, backend._window_array(backend.Window.Hann, W))[1]} frames "
                        "(a short estimate: fewer than 128 frames)", unit="ms per call, host arrays in and out")
    for name, prec in (("f64_route_auto", "auto"), ("f32_kernels_forced", "f32")):
        ts = []
        for i in range(9):
            t0 = time.perf_counter()
            tf, coh = backend.welch_transfer_function(y, x, FS, W, "H1", precision=prec)
            if i >= 2:
                ts.append((time.perf_counter() - t0) * 1e3)
        ent[name] = dict(ms_median=float(np.median(ts)), ms_min=float(np.min(ts)),
                         parity_rel_max_vs_oracle=max(orc.rel_max(tf[band], rt[band]), orc.rel_max(coh[band], rc[band])))
    return ent


def main():
    args = parse_args()
    args.cpu_bounded = False
    if args.gpus < 1:
        print("[bench] --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # nothing has touched the GPU in this process
    from dsptoolbox_amd._build import build_library
    from dsptoolbox_amd._lib import Context

    build_library()
    dist = Dist(args.gpus)
    ctx = Context(None)  # device = LOCAL_RANK (modulo the visible device count)
    rccl = setup_rccl(ctx, dist, required=(args.scaling == "strong"))
    strong = args.scaling == "strong" and dist.world > 1
    out, made, wall = measure(args, ctx, dist, args.workload, args.steps, args.warmup, strong, rccl)
    if dist.rank != 0:
        dist.finish()
        return
    if "roofline" not in out:
        print(json.dumps(out), flush=True)
        dist.finish()
        return
    if dist.world > 1:
        out["bcast"] = "rccl" if rccl else "host"
        out["host_exchange"] = "dsptoolbox_amd.rendezvous.TcpExchange (no torch in this process: " + \
                               ("true" if "torch" not in sys.modules else "false") + ")"
        if rccl:
            n = C.c_int(0)
            ctx.check(ctx.lib.ds_comm_count(ctx.handle, C.byref(n)), "ds_comm_count")
            out["rccl_ranks"] = int(n.value)  # what RCCL itself says (ncclCommCount of the library's communicator)
        if RCCL_ERROR:
            out["rccl_error"] = RCCL_ERROR
        if strong:
            out["result_gather"] = "rccl all-gather per step" if rccl else "none"
    if dist.world == 1 and args.predict_ranks > 1:
        # what one rank of an N-GPU strong-scaling job would run, timed on this GPU (no collective
        # here: the all-gather of 24 KB result slices and the one-off broadcast are extra)
        R = args.predict_ranks
        maker = dict(welch_h1=welch_h1, welch_h1_1024=lambda a, c, d, sh, r: welch_h1(a, c, d, sh, r, W=1024),
                     fir_bank=fir_bank, csm=csm, deconv=deconv)[args.workload]
        s_step = maker(args, ctx, dist, (0, R), False)[0]
        for _ in range(max(3, args.warmup // 2)):
            s_step()
        s_wall, s_ev, _ = timed_steps(ctx, dist, s_step, args.steps, False, None)
        out["shard_prediction"] = dict(
            ranks=R, shard="rank 0 of shard_range over " + {"welch_h1": "64 output channels (+ the sweep)",
                                                            "welch_h1_1024": "64 output channels (+ the sweep)",
                                                            "fir_bank": "32 bands", "csm": "513 bins",
                                                            "deconv": "1024 items"}[args.workload],
            ms_full=out["ms_per_step"], ms_shard=s_wall * 1e3 / args.steps, implied_speedup=wall / s_wall,
            note="single-GPU timing of the per-rank shape; excludes the per-step all-gather")
    cpu_legs = made[5]
    if cpu_legs and not args.no_cpu_baseline and dist.world == 1:
        out["cpu_baseline"] = cpu_legs[0]()
        out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        if len(cpu_legs) > 1:
            out["cpu_baseline_batched"] = cpu_legs[1]()
    if args.workload == "welch_h1" and dist.world == 1 and not args.no_workloads and not args.no_cpu_baseline:
        # the other BASELINE configs, driver-timed inside the default line
        import gc
        del made, cpu_legs
        gc.collect()
        out["workloads"] = {}
        t_all = time.perf_counter()
        for name in ("welch_h1_1024", "fir_bank", "csm", "deconv"):
            t0 = time.perf_counter()
            out["workloads"][name] = workload_entry(args, ctx, dist, name)
            out["workloads"][name]["wall_s_including_setup_and_cpu_leg"] = time.perf_counter() - t0
        # config 2 with detrend on AND off (the line itself is --detrend 1: the reference's default)
        keep = args.detrend
        args.detrend = 0 if keep else 1
        out["workloads"]["welch_h1_detrend_" + ("off" if keep else "on")] = workload_entry(args, ctx, dist, "welch_h1")
        args.detrend = keep
        out["workloads"]["short_estimate_api"] = short_estimate_entry()
        out["workloads_wall_s"] = time.perf_counter() - t_all
    print(json.dumps(out), flush=True)
    dist.finish()


if __name__ == "__main__":
    main()
