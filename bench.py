#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X spectral hot path.

    python bench.py --gpus N --steps K --warmup W [--workload welch_h1|welch_h1_1024|fir_bank|csm|deconv]

Default workload (BASELINE.json configs[1], the one the metric is quoted on):
64-channel Welch H1 transfer-function estimation, one sweep input channel,
2^20 samples per channel, nfft 4096, Hann, 50 % overlap.  One step = one
ds_welch_tf_dev call over inputs that are already resident in HBM.
N > 1: one process per GPU (torch.distributed.run), every rank owns an
independent 64-channel batch (weak scaling, no data-path collective); the
shared sweep channel is broadcast once over RCCL/xGMI before the timed region.

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

FS = 48000


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="welch_h1",
                    choices=["welch_h1", "welch_h1_1024", "fir_bank", "csm", "deconv"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-channels", type=int, default=64,
                    help="output channels of the bounded CPU-baseline sample")
    ap.add_argument("--detrend", type=int, default=1)
    return ap.parse_args()


# ---------------------------------------------------------------------------
class Dist:
    """torch.distributed only as plumbing: rendezvous, barrier, max over ranks."""

    def __init__(self, n_gpus: int):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.torch = None
        try:
            import torch
            self.torch = torch
        except Exception:  # pragma: no cover
            pass
        self.backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # gloo: rehearsal on one GPU
        if self.world > 1:
            import torch.distributed as dist
            if self.backend == "nccl":
                self.torch.cuda.set_device(self.local_rank)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=self.backend)
            self.dist = dist
        assert self.world == n_gpus or self.world == 1, (self.world, n_gpus)

    def barrier_sync(self, ctx):
        ctx.sync()
        if self.torch is not None and self.torch.cuda.is_available():
            self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, v: float) -> float:
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64,
                              device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def bcast_bytes(self, b: bytes, n: int) -> bytes:
        if self.world == 1:
            return b
        box = [b if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    comm_hung = False

    def finish(self):
        if self.world > 1:
            try:
                self.dist.barrier()
                self.dist.destroy_process_group()
            except Exception:  # pragma: no cover - teardown only
                pass
        if self.comm_hung:  # a daemon thread is stuck inside the communicator set-up
            sys.stdout.flush()
            os._exit(0)

    def all_ok(self, ok: bool) -> bool:
        if self.world == 1:
            return ok
        t = self.torch.tensor([1.0 if ok else 0.0], dtype=self.torch.float64,
                              device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)


def setup_rccl(ctx, dist: Dist):
    """Library-level RCCL communicator (ds_comm_*) used for the sweep broadcast.  Returns
    False (-> host broadcast + upload) for one rank, or if RCCL cannot be set up, e.g. a
    gloo rehearsal with several ranks on one GPU."""
    if dist.world == 1:
        return False
    if dist.backend != "nccl":
        return False
    if os.environ.get("BENCH_BCAST", "rccl") != "rccl":  # BENCH_BCAST=torch: torch.distributed broadcast
        return False
    ident = C.create_string_buffer(128)
    ok = True
    if dist.rank == 0:
        ok = ctx.lib.ds_comm_unique_id(ident) == 0
    raw = dist.bcast_bytes(ident.raw, 128)
    if not dist.all_ok(ok):
        return False
    # ncclCommInitRank is collective: guard it with a watchdog so that a rendezvous problem
    # degrades to the host broadcast instead of hanging the whole multi-GPU run
    import threading
    res = {}

    def init():
        res["rc"] = ctx.lib.ds_comm_init(ctx.handle, dist.world, dist.rank, raw)

    th = threading.Thread(target=init, daemon=True)
    th.start()
    th.join(timeout=float(os.environ.get("BENCH_RCCL_TIMEOUT", "90")))
    if th.is_alive():
        dist.comm_hung = True  # leave through os._exit at the end
        print(f"[bench] rank {dist.rank}: ds_comm_init did not return; using torch.distributed", file=sys.stderr,
              flush=True)
    return dist.all_ok(res.get("rc", -1) == 0)


# ---------------------------------------------------------------------------
def welch_h1(args, ctx, dist, W=4096):
    from dsptoolbox_amd import backend
    from dsptoolbox_amd._lib import DeviceBuffer
    from dsptoolbox_amd.generators import exponential_sweep, sweep_and_responses
    from dsptoolbox_amd.standard.enums import SpectrumScaling, Window

    n, n_cy = 2**20, 64
    # per-rank independent batch (different response / noise seeds), shared sweep
    x, y = sweep_and_responses(n, n_cy, FS)
    if dist.rank > 0:
        rng = np.random.default_rng(9000 + dist.rank)
        y = y[:, rng.permutation(n_cy)] * (1.0 + 0.01 * dist.rank)
    window = backend._window_array(Window.Hann, W)
    hop, n_frames = backend._welch_framing(n, W, 50, window)
    amp, norm_scale, factor, phys = backend._finish_params(SpectrumScaling.FFTBackward, W, FS, window)
    d_y = DeviceBuffer.from_array(ctx, backend._planar_f32(y))
    d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
    xp = backend._planar_f32(x)
    d_x = DeviceBuffer(ctx, xp.nbytes)
    bcast_ms = None
    if setup_rccl(ctx, dist):
        if dist.rank == 0:
            ctx.upload(d_x.ptr, xp)
        dist.barrier_sync(ctx)
        t0 = time.perf_counter()
        ctx.check(ctx.lib.ds_bcast(ctx.handle, C.c_void_p(d_x.ptr), xp.nbytes, 0), "ds_bcast")
        ctx.sync()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    else:
        if dist.world > 1:  # no RCCL: broadcast on the host, then upload
            from dsptoolbox_amd.distributed import broadcast_array
            xp = broadcast_array(xp if dist.rank == 0 else None, src=0)
        ctx.upload(d_x.ptr, xp)
    B = W // 2 + 1
    d_tf = DeviceBuffer(ctx, B * n_cy * 8)
    d_coh = DeviceBuffer(ctx, B * n_cy * 4)

    def step():
        ctx.check(ctx.lib.ds_welch_tf_dev(
            ctx.handle, C.c_void_p(d_x.ptr), 1, n, C.c_void_p(d_y.ptr), n_cy, n, n, W, hop, n_frames,
            C.c_void_p(d_w.ptr), int(args.detrend), 0, 1, amp, norm_scale, factor, phys,
            C.c_void_p(d_tf.ptr), C.c_void_p(d_coh.ptr)), "ds_welch_tf_dev")

    samples_per_step = (n_cy + 1) * n
    alg_bytes = (n_cy + 1) * n * 4 + B * n_cy * 8 + B * n_cy * 4
    info = dict(
        workload=f"welch_h1: 64 output ch + 1 sweep input ch x 2^20 samples, nfft {W}, Hann, 50% overlap"
                 + (", detrend" if args.detrend else ""),
        channels=n_cy, samples_per_channel=n, nfft=W, overlap_percent=50, frames=n_frames,
        parallelism=f"channel-batch x{dist.world}")

    def verify():
        tf = d_tf.to_array((B, n_cy), np.complex64)
        assert np.all(np.isfinite(tf[1:]))
        return tf, d_coh.to_array((B, n_cy), np.float32)

    def cpu_baseline():
        from oracle import dsp_oracle as orc
        cc = min(args.cpu_channels, n_cy)
        reps = 2  # ~14 s of single-core work for the full 64 channels
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

    return step, samples_per_step, alg_bytes, "hbm", info, cpu_baseline, bcast_ms, \
        ("welch4096_main", "welch1024_main", "welch_yacc")


def fir_bank(args, ctx, dist):
    from dsptoolbox_amd import backend
    from dsptoolbox_amd._lib import DeviceBuffer
    from dsptoolbox_amd.generators import fir_bank_taps

    n, n_ch, K, T = 2**22, 8, 32, 4097
    x = np.random.default_rng(3 + dist.rank).standard_normal((n, n_ch)) * 0.1
    taps = fir_bank_taps(K, T, FS).astype(np.float32)
    d_x = DeviceBuffer.from_array(ctx, backend._planar_f32(x))
    d_t = DeviceBuffer.from_array(ctx, taps)
    d_y = DeviceBuffer(ctx, K * n_ch * n * 4)

    def step():
        ctx.check(ctx.lib.ds_fir_ola_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n,
                                         C.c_void_p(d_t.ptr), K, T, backend.DS_FB_PARALLEL,
                                         C.c_void_p(d_y.ptr), n), "ds_fir_ola_dev")

    alg_bytes = n_ch * n * 4 + K * T * 4 + K * n_ch * n * 4
    info = dict(workload="fir_bank: FilterBank Parallel, 32 x 4097-tap FIR over 8 ch x 2^22",
                channels=n_ch, samples_per_channel=n, bands=K, taps=T,
                parallelism=f"signal-batch x{dist.world}")

    def cpu_baseline():
        from oracle import dsp_oracle as orc
        nb = K  # the reference loop: one oaconvolve per band (~0.5 s each)
        err = 0.0
        t0 = time.perf_counter()
        for k in range(nb):
            ref = orc.lfilter_fir(taps[k].astype(np.float64), x)
            if k % 8 == 0:  # parity on every 8th band (the download is outside what is measured)
                t1 = time.perf_counter()
                got = np.empty((n_ch, n), dtype=np.float32)
                ctx.download(d_y.ptr + 4 * k * n_ch * n, got)
                err = max(err, orc.rel_max(got.T, ref))
                t0 += time.perf_counter() - t1
        dt = time.perf_counter() - t0
        return dict(value=n_ch * n / dt / 1e6, unit="Msamples/s", cores=1, kind="port",
                    sample=f"oracle.lfilter_fir (scipy oaconvolve), all {K} bands, {dt:.1f} s",
                    parity_rel_max_vs_gpu=err)

    return step, n_ch * n, alg_bytes, "hbm", info, cpu_baseline, None, ("fir",)


def csm(args, ctx, dist):
    from dsptoolbox_amd import backend
    from dsptoolbox_amd._lib import DeviceBuffer
    from dsptoolbox_amd.generators import mic_array_noise
    from dsptoolbox_amd.standard.enums import SpectrumScaling, Window

    n, n_ch, W = 512000, 64, 1024
    x = mic_array_noise(n, n_ch, 4 + dist.rank)
    window = backend._window_array(Window.Hann, W)
    hop, n_frames = backend._welch_framing(n, W, 50, window)
    amp, norm_scale, factor, phys = backend._finish_params(SpectrumScaling.FFTBackward, W, FS, window)
    d_x = DeviceBuffer.from_array(ctx, backend._planar_f32(x))
    d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
    B = W // 2 + 1
    d_c = DeviceBuffer(ctx, B * n_ch * n_ch * 8)

    def step():
        ctx.check(ctx.lib.ds_csm_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n, W, hop, n_frames,
                                     C.c_void_p(d_w.ptr), 1, 0, amp, norm_scale, factor, phys,
                                     C.c_void_p(d_c.ptr)), "ds_csm_dev")

    flops = B * n_ch * n_ch * n_frames * 8.0
    info = dict(workload="csm: 64-mic Welch cross-spectral matrix, nfft 1024, 1000 frames",
                channels=n_ch, samples_per_channel=n, nfft=W, frames=n_frames,
                gemm_flops="513*64*64*1000*8 (full Hermitian count)",
                parallelism=f"signal-batch x{dist.world}")

    def cpu_baseline():
        from oracle import dsp_oracle as orc
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            f, ref = orc.csm_welch_batched(x, FS, W, "hann", 50, True, "FFTBackward", workers=-1)
        dt = (time.perf_counter() - t0) / reps
        got = d_c.to_array((B, n_ch, n_ch), np.complex64)
        return dict(value=n_ch * n / dt / 1e6, unit="Msamples/s", cores=os.cpu_count(), kind="port",
                    sample=f"oracle.csm_welch_batched (batched restatement, scipy.fft workers=-1), all 64 "
                           f"mics, {reps} passes of {dt:.1f} s; the reference's 2080-pair loop takes ~100 s",
                    parity_rel_max_vs_gpu=orc.rel_max(got[1:], ref[1:]))

    return step, n_ch * n, flops, "mfma", info, cpu_baseline, None, ("csm_gemm",)


def deconv(args, ctx, dist):
    from dsptoolbox_amd import backend
    from dsptoolbox_amd._lib import DeviceBuffer
    from dsptoolbox_amd.generators import exponential_sweep

    n, items, n_ch = 8192, 1024 // max(dist.world, 1) * 1, 2
    x = exponential_sweep(n, FS)
    rng = np.random.default_rng(5000 + dist.rank)
    y = rng.standard_normal((items, n_ch, n)).astype(np.float32) * 0.1
    r = (rng.standard_normal(n // 2 + 1) + 1j * rng.standard_normal(n // 2 + 1)).astype(np.complex64)
    d_y = DeviceBuffer.from_array(ctx, y)
    d_r = DeviceBuffer.from_array(ctx, r)
    d_o = DeviceBuffer(ctx, y.nbytes)

    def step():
        ctx.check(ctx.lib.ds_deconv_dev(ctx.handle, C.c_void_p(d_y.ptr), items, n_ch, n, n, n,
                                        C.c_void_p(d_r.ptr), 0, n, n, C.c_void_p(d_o.ptr)),
                  "ds_deconv_dev")

    alg_bytes = 2 * y.nbytes + r.nbytes
    info = dict(workload="deconv: stereo spectral deconvolutions n=8192 against a shared inverse sweep",
                items=items, channels=n_ch, samples_per_channel=n,
                parallelism=f"item-shard x{dist.world}")
    def cpu_baseline():
        # the reference's per-item path (_transfer_functions.py:19-42): rfft, multiply, irfft
        yy = y.astype(np.float64)
        rr = r.astype(np.complex128)
        reps = 80  # ~0.1 s per pass
        t0 = time.perf_counter()
        for _ in range(reps):
            for i in range(items):
                ref = np.fft.irfft(np.fft.rfft(yy[i], n=n, axis=-1) * rr, n=n, axis=-1)
        dt = (time.perf_counter() - t0) / reps
        got = d_o.to_array((items, n_ch, n), np.float32)[-1]
        den = float(np.max(np.abs(ref)))
        return dict(value=items * n_ch * n / dt / 1e6, unit="Msamples/s", cores=1, kind="port",
                    sample=f"numpy rfft * R -> irfft per item (reference loop), all {items} items, "
                           f"{reps} passes of {dt:.1f} s",
                    parity_rel_max_vs_gpu=float(np.max(np.abs(got - ref))) / den)

    return step, items * n_ch * n, alg_bytes, "hbm", info, cpu_baseline, None, ("deconv",)


def pmc_traffic(workload: str, kernel_hint: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of
    this same command (profiles/rNN_<workload>_rocprofv3_summary.txt; FETCH_SIZE and WRITE_SIZE
    are KiB, collected in separate --pmc passes; on gfx950 FETCH_SIZE counts half of a coalesced
    streaming read -- MI355X_MICROARCH.md, HBM section -- hence the factor 2).  None if absent."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_rocprofv3_summary.txt")))
    if not files:
        return None, None
    text = open(files[-1]).read()
    sect = text.split("== PMC", 1)[-1]
    blocks = re.split(r"\n(?=\S)", sect)
    for blk in blocks:
        head = blk.splitlines()[0] if blk.strip() else ""
        if kernel_hint not in head:
            continue
        f = re.search(r"FETCH_SIZE\s+([0-9.]+)", blk)
        w = re.search(r"WRITE_SIZE\s+([0-9.]+)", blk)
        if f and w:
            return (2.0 * float(f.group(1)) + float(w.group(1))) * 1024.0, os.path.basename(files[-1])
    return None, None


# ---------------------------------------------------------------------------
def main():
    args = parse_args()
    from dsptoolbox_amd._build import build_library
    from dsptoolbox_amd._lib import Context

    build_library()
    dist = Dist(args.gpus)
    ctx = Context(None)  # device = LOCAL_RANK (modulo the visible device count)
    maker = dict(welch_h1=welch_h1, welch_h1_1024=lambda a, c, d: welch_h1(a, c, d, W=1024),
                 fir_bank=fir_bank, csm=csm, deconv=deconv)[args.workload]
    step, units, alg, bound, info, cpu_baseline, bcast_ms, dominant = maker(args, ctx, dist)

    # Warm-up; its last step is bracketed kernel by kernel (HIP events on the library's stream) to
    # find the dominant kernel and the per-kernel breakdown.  An event pair costs ~3 us of stream
    # time, so in the timed region only the dominant kernel is bracketed.
    events = not os.environ.get("BENCH_NO_KERNEL_EVENTS")
    prof_all = {}
    for i in range(args.warmup):
        if events and i == args.warmup - 1:
            ctx.sync()
            ctx.profile_enable(True)
            ctx.profile_report()
            step()
            prof_all = ctx.profile_report()
            ctx.profile_enable(False)
        else:
            step()
    dom = next((k for k in dominant if k in prof_all), None)
    if dom is None and prof_all:
        dom = max(prof_all, key=lambda k: prof_all[k][0])
    dist.barrier_sync(ctx)
    ctx.profile_only(dom)  # None (no instrumented warm-up step): every kernel
    ctx.profile_enable(events)
    ctx.profile_report()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_stop()
    dist.barrier_sync(ctx)
    wall = time.perf_counter() - t0
    prof = ctx.profile_report()
    ctx.profile_enable(False)
    ctx.profile_only(None)
    wall = dist.max_over_ranks(wall)

    if dist.rank != 0:
        dist.finish()
        return
    ms_per_step = wall * 1e3 / args.steps
    value = units * dist.world / (wall / args.steps) / 1e6
    if not prof:  # BENCH_NO_KERNEL_EVENTS=1: step time without the per-kernel event markers (dev)
        print(json.dumps({"value": value, "ms_per_step": ms_per_step, "step_event_ms": ev_ms / args.steps,
                          "note": "no per-kernel events: no roofline"}), flush=True)
        dist.finish()
        return
    if dom is None or dom not in prof:
        dom = next((k for k in dominant if k in prof), max(prof, key=lambda k: prof[k][0]))
    dom_ms = prof[dom][0] / prof[dom][1]
    launches_per_step = prof[dom][1] / args.steps
    if bound == "hbm":
        achieved = alg / launches_per_step / (dom_ms * 1e-3) / 1e9
        roof = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=None)
    else:
        achieved = alg / launches_per_step / (dom_ms * 1e-3) / 1e12
        roof = dict(bound="mfma", achieved=achieved, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=achieved / MFMA_F32_PEAK_TFLOPS, traffic=None)
    hints = {"welch4096_main": ("welch4096::k_y<",), "welch1024_main": ("welch1k::k_y<",), "welch_yacc": ("k_yacc",), "fir": ("fir16k::k_fir<true>", "fir16k::k_fir", "k_fir<"),
             "csm_gemm": ("k_csm_gemm",), "deconv": ("k_deconv",)}.get(dom, (dom,))
    traffic, src = None, None
    for hint in hints:
        traffic, src = pmc_traffic(args.workload, hint)
        if traffic is not None:
            break
    roof["traffic"] = traffic
    if src:
        roof["traffic_source"] = src + " (2*FETCH_SIZE + WRITE_SIZE) KiB"
    roof["kernel"] = dom
    roof["kernel_avg_ms"] = dom_ms
    roof["algorithmic_per_launch"] = alg / launches_per_step
    out = {
        "metric": "Msamples/s + GB/s vs HBM roofline, 64ch Welch H1 nfft=4096 @1/2/4/8 GPU"
                  if args.workload == "welch_h1" else f"Msamples/s ({args.workload})",
        "value": value, "unit": "Msamples/s", "n_gpus": dist.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": info, "roofline": roof,
        "step_event_ms": ev_ms / args.steps,
        "kernels_ms_per_step": ({k: v[0] for k, v in prof_all.items()} if prof_all
                                else {k: v[0] / args.steps for k, v in prof.items()}),
        "kernels_ms_per_step_source": ("last warm-up step, every kernel bracketed" if prof_all
                                       else "timed region"),
        "whole_step_gbs": (alg / (ev_ms / args.steps * 1e-3) / 1e9) if bound == "hbm" else None,
    }
    if bcast_ms is not None:
        out["rccl_bcast_ms"] = bcast_ms
    if cpu_baseline is not None and not args.no_cpu_baseline and dist.world == 1:
        out["cpu_baseline"] = cpu_baseline()
        out["cpu_baseline"]["host_cpus"] = os.cpu_count()
    print(json.dumps(out), flush=True)
    dist.finish()


if __name__ == "__main__":
    main()
