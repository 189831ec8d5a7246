#!/usr/bin/env python3
"""Headline benchmark: FFT-VACF lag-points/s on synthetic random velocities.

    python bench.py [--gpus N --steps K --warmup W]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU (RCCL).  Workload at every N: BASELINE.json configs[2]'s shape,
10 000 frames x 100 000 atoms x 3 float64 PER GPU (weak scaling: the atom axis is
the sharded unit, each rank owns a contiguous block of atoms), device-resident
before the timed region.  A step = one pass of the hot path over the rank's
block (ta_vacf_fft_dev: power-spectrum accumulation + one inverse transform)
followed, for N > 1, by the single all-reduce of the (n_frames,) lag sums.

Prints ONE JSON line (rank 0).  `value` = total frames x atoms processed per
second over all ranks; `roofline` prices the dominant kernel (k_fft_accum)
against the 8 TB/s HBM roof with its algorithmic bytes (n_frames*n_atoms*dim*8 per
launch) and its hipEvent-measured duration; `cpu_baseline` is the NumPy oracle
(per-atom loop + numpy.fft, like the reference) on one host core over an atom
subsample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--atoms", type=int, default=100000, help="atoms PER GPU")
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--mode", default="fft", choices=["fft", "direct", "helfand"])
    ap.add_argument("--by-particle", action="store_true",
                    help="also materialise vacf_by_particle (secondary number)")
    ap.add_argument("--float32", action="store_true",
                    help="direct / helfand modes: float32 products and block sums (configs[4])")
    ap.add_argument("--helfand-fft", action="store_true",
                    help="--mode helfand: the O(T log T) option (lag sums as S1 - 2 S2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-atoms", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(args, T, D):
    """NumPy oracle (reference control flow: per-atom loop, tidynamics-style FFT) on
    ONE core over an atom subsample; linear in the atom count."""
    import numpy as np

    from oracle import numpy_oracle as orc

    all_cpus = None
    try:
        all_cpus = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {sorted(all_cpus)[0]})
    except Exception:
        pass
    if args.mode == "fft":
        a = args.cpu_sample_atoms or max(8, int(8.0e7 // T))  # ~10-15 s of CPU work
        v = orc.synthetic_velocities(T, a, D, seed=20250824 + 3)
        t0 = time.perf_counter()
        orc.vacf_fft(v)
        dt = time.perf_counter() - t0
        what = f"oracle.numpy_oracle.vacf_fft on {T} frames x {a} atoms x {D} (atom subsample of the workload)"
    elif args.mode == "direct":
        a = args.cpu_sample_atoms or max(2, int(2.0e9 // (T * T)))
        v = orc.synthetic_velocities(T, a, D, seed=20250824 + 4)
        t0 = time.perf_counter()
        orc.vacf_windowed(v)
        dt = time.perf_counter() - t0
        what = f"oracle.numpy_oracle.vacf_windowed on {T} x {a} x {D} (atom subsample)"
    else:
        a = args.cpu_sample_atoms or max(2, int(1.0e9 // (T * T)))
        v, x, m, vol = orc.synthetic_helfand(T, a, D, seed=20250824 + 5)
        t0 = time.perf_counter()
        orc.helfand(v, x, m, vol)
        dt = time.perf_counter() - t0
        what = f"oracle.numpy_oracle.helfand on {T} x {a} x {D} (atom subsample)"
    out = {"value": T * a / dt, "unit": "lag-points/s", "cores": 1, "kind": "port",
           "sample": what, "seconds": round(dt, 2)}
    # second line: the plain-C oracle (OpenMP, own radix-2 FFT / slab loops) on every host core
    # over the same subsample -- the "best CPU" figure next to the reference-like NumPy one
    try:
        from oracle import c_oracle

        if all_cpus:
            os.sched_setaffinity(0, all_cpus)
        n = len(all_cpus) if all_cpus else (os.cpu_count() or 1)
        t0 = time.perf_counter()
        if args.mode == "fft":
            c_oracle.vacf_fft_lagsum(v, n_threads=n)
        elif args.mode == "direct":
            c_oracle.vacf_windowed(v, n_threads=n)
        else:
            c_oracle.helfand(v, x, m, vol, n_threads=n)
        dtc = time.perf_counter() - t0
        out["all_cores"] = {"value": T * a / dtc, "unit": "lag-points/s", "cores": n,
                            "kind": "port", "impl": "oracle/c (OpenMP)", "seconds": round(dtc, 2)}
    except Exception as e:  # the C oracle is optional test infrastructure
        out["all_cores"] = {"error": str(e)[:200]}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with python -m torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from transport_analysis_amd import _lib
    from transport_analysis_amd.dist import reduce_lagsum

    T, A, D = args.frames, args.atoms, args.dim
    ctx = _lib.Context(local_rank)
    if args.float32:
        if args.mode == "fft":
            raise SystemExit("--float32 applies to --mode direct / helfand")
        ctx.set_option("direct_f32", 1)
    if args.helfand_fft:
        if args.mode != "helfand" or args.float32:
            raise SystemExit("--helfand-fft applies to --mode helfand without --float32")
        ctx.set_option("helfand_fft", 1)
    gen = torch.Generator(device=dev)
    gen.manual_seed(20250824 + 3 + 1000 * rank)
    vel = torch.randn((T, A, D), dtype=torch.float64, device=dev, generator=gen)
    pos = masses = None
    if args.mode == "helfand":
        pos = 30.0 + 0.002 * torch.cumsum(vel, dim=0)
        masses = torch.tensor([15.999, 1.008, 1.008], dtype=torch.float64, device=dev).repeat(
            (A + 2) // 3)[:A].contiguous()
    lagsum = torch.zeros(T, dtype=torch.float64, device=dev)
    bp = torch.empty((T, A), dtype=torch.float64, device=dev) if args.by_particle else None
    stream = torch.cuda.current_stream().cuda_stream
    a_total = A * world

    def step():
        d_bp = bp.data_ptr() if bp is not None else 0
        if args.mode == "fft":
            ctx.vacf_fft_dev(vel.data_ptr(), T, A, D, A * D, lagsum.data_ptr(), d_bp, A, stream)
        elif args.mode == "direct":
            ctx.vacf_direct_dev(vel.data_ptr(), T, A, D, A * D, lagsum.data_ptr(), d_bp, A, stream)
        else:
            ctx.helfand_msd_dev(vel.data_ptr(), pos.data_ptr(), masses.data_ptr(), T, A, D, A * D,
                                1.0, lagsum.data_ptr(), d_bp, A, stream)
        return reduce_lagsum(lagsum, a_total)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ts = step()
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = step()
        # hipEvent pair recorded by the library around the dominant kernel on this stream
        # (reading it waits for that launch only; it is inside the timed region on purpose)
        kernel_ms.append(ctx.last_timing()[1])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # cheap validity check of the timed result: a few lags recomputed with torch ops
    errs = []
    if args.mode in ("fft", "direct"):
        loc = lagsum if world == 1 else None
        if loc is None:  # lagsum was all-reduced in place: recompute the local part
            loc = torch.zeros(T, dtype=torch.float64, device=dev)
            ctx.vacf_fft_dev(vel.data_ptr(), T, A, D, A * D, loc.data_ptr(), 0, A, stream)
        scale = float((vel * vel).sum().item()) / T
        for k in (0, 1, T // 2, T - 1):
            ref = float((vel[: T - k] * vel[k:]).sum().item()) / (T - k)
            errs.append(abs(float(loc[k].item()) - ref) / scale)
    torch.cuda.synchronize()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ms_per_step = elapsed / args.steps * 1e3
    main_ms = sum(kernel_ms) / len(kernel_ms)
    bytes_algo = T * A * D * 8 * (2 if args.mode == "helfand" else 1)
    achieved = bytes_algo / (main_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            rec = json.load(open(tfile))
            key = f"{args.mode}_{T}x{A}x{D}"
            if key in rec and not args.by_particle:
                traffic = rec[key]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
    out = {
        "metric": "VACF lag-points/sec (n_frames x n_atoms / s)",
        "value": T * a_total / (elapsed / args.steps),
        "unit": "lag-points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (f64 inputs and accumulators)" if args.float32 else "f64",
        "data": "synthetic",
        "config": {
            "workload": {"fft": "FFT VACF", "direct": "windowed (direct) VACF",
                         "helfand": "Helfand MSD (FFT option)" if args.helfand_fft else "Helfand MSD"}[args.mode]
                        + (" with the by-particle array" if args.by_particle else " timeseries")
                        + f", {T} frames x {A} atoms x {D} float64 per GPU"
                        + (" (BASELINE configs[2] shape)" if (T, A, D, args.mode) == (10000, 100000, 3, "fft") else ""),
            "n_frames": T, "n_atoms_per_gpu": A, "dim": D, "mode": args.mode,
            "by_particle": bool(args.by_particle), "sharding": f"atoms x{world}",
            "fft_plan": _lib.fft_plan_info(T),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "kernel": ("k_fft_accum" if T <= 10240 or args.by_particle else "k_fft_accum_long")
                      if args.mode == "fft" else "k_direct",
            "kernel_ms": main_ms, "algorithmic_bytes_per_launch": bytes_algo,
        },
        "check": {"max_scale_rel_err_vs_torch_lags": max(errs) if errs else None},
    }
    if args.mode != "fft" and not args.helfand_fft:
        # the direct correlators are bound by the vector FP issue rate:
        # windowed VACF 2*D*A*T(T+1)/2 flop, Helfand 3*D*A*T(T-1)/2; HBM is touched once
        flops = (2.0 * D * A * T * (T + 1) / 2) if args.mode == "direct" else (3.0 * D * A * T * (T - 1) / 2)
        peak = 157.3 if args.float32 else 78.6
        tf = flops / (main_ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "valu", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                           "frac": tf / peak, "traffic": None, "kernel": "k_direct", "kernel_ms": main_ms,
                           "algorithmic_flops_per_launch": flops,
                           "hbm_GBps_for_reference": achieved}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, T, D)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
