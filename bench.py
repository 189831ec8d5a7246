#!/usr/bin/env python3
"""Headline benchmark: FFT-VACF lag-points/s on synthetic random velocities.

    python bench.py [--gpus N --steps K --warmup W] [--scaling weak|strong]

N > 1 runs one rank per GPU (RCCL): either the driver starts the ranks,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
or a bare `python bench.py --gpus N` starts that very command itself as a CHILD process (before
this process has imported torch or touched a GPU), lets rank 0's JSON line through and exits
with the child's status.  Workload: BASELINE.json configs[2]'s tensor, 10 000 frames x
100 000 atoms x 3 float64 -- per GPU with --scaling weak (the default: the atom axis is the
sharded unit, per-GPU work fixed), in total with --scaling strong (configs[2] exactly: 100 000
atoms over the N GPUs).  Every rank materialises ITS column block of ONE synthetic tensor with
the library's stateless counter-based generator (ta_stage_synth; oracle/synth.py is its NumPy
twin) straight into the library's device slab: the input is resident in HBM, in the layout the
staging path leaves it in, before the timed region starts.  A step = one pass of the hot path
over the rank's block (ta_vacf_fft_staged: power-spectrum accumulation + one inverse transform)
followed, for N > 1, by the single all-reduce of the (n_frames,) lag sums (device tensor in,
device tensor out).

Prints ONE JSON line (rank 0).  `value` = total frames x atoms processed per second over all
ranks (from the wall time of exactly K steps between two fences, max over ranks).
`roofline` prices the dominant kernel (k_wsplit_accum) against the 8 TB/s HBM roof with its
algorithmic bytes (n_frames*n_atoms*dim*8 per launch) and the median of its K hipEvent-measured
durations (recorded by the library on the launch stream, read AFTER the timed region), and
carries the FP64 vector co-roof beside it.  `cpu_baseline` is the NumPy oracle (per-atom loop
+ numpy.fft, the reference's control flow) on one host core over an atom block of the same
tensor.  `other_configs` (N = 1) are short driver-timed runs of the other BASELINE configs; their
`roofline.frac` is the WHOLE path's algorithmic bytes (or flops) over the whole call's device
time, with the per-kernel split (`kernels`, from the library's event timeline of one extra,
untimed step) beside it; the float64 lag sums of the O(T^2) correlators (configs[3], configs[4])
run on the FP64 matrix cores and are priced against the matrix peak (`bound: "mfma"`).  `staging` prices what the timed region leaves out: the on-device
transposition of frame-major frames into the pair-major slab (k_relayout) and a complete
ta_vacf_fft_dev call on a frame-major device tensor.  `host_path_by_particle` runs the drop-in
CLASS end to end (frames staged through pinned memory, result array in pinned memory).
N > 1 (and N = 1 under torchrun with TA_BENCH_FORCE_DIST=1): `config.rank_devices` lists the GPU
every rank ran on (the run fails unless they are N distinct devices) and `reduce_us` is the
all-reduce alone.
`--single-process` (not under torchrun) runs the N GPUs from THIS process through the library's
own fan-out (ta_group: one context per device, the reduce inside the call -- RCCL for distinct
devices); a step is then one host-facing ta_group_vacf_fft call, the (n_frames,) result on the
host included.  `--devices 0,0` picks the members explicitly (two members on one GPU: what a
one-GPU box can rehearse).  `device_ms.effective_mhz` is the clock the headline kernel ran at,
measured inside a stamped build of it (ta_clock_probe) right after the timed region;
`roofline.model_ceiling` is what this transform structure can reach under the package power
limit.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # vector FP64
FP32_PEAK_TFLOPS = 157.3
FP64_MFMA_PEAK_TFLOPS = 78.6  # matrix FP64 (v_mfma_f64_16x16x4_f64: 32 flop / clock / SIMD, as the vector pipe)
FP32_MFMA_PEAK_TFLOPS = 157.3  # matrix FP32 (v_mfma_f32_16x16x4_f32: 64 flop / clock / SIMD; 155 measured, MI355X guide)
SEED = 20250824


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--atoms", type=int, default=100000,
                    help="atoms per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--mode", default="fft", choices=["fft", "direct", "helfand"])
    ap.add_argument("--by-particle", action="store_true",
                    help="also materialise vacf_by_particle (the reference's default output)")
    ap.add_argument("--float32", action="store_true",
                    help="direct / helfand modes: float32 products and block sums (configs[4])")
    ap.add_argument("--slab32", action="store_true",
                    help="--mode fft: float32 device slabs (the tensor rounded once to float32; MDAnalysis' dtype), "
                         "float64 arithmetic")
    ap.add_argument("--helfand-fft", action="store_true",
                    help="--mode helfand: the O(T log T) option (lag sums as S1 - 2 S2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-kernel-split", action="store_true",
                    help="skip the extra untimed step that records the per-kernel event timeline")
    ap.add_argument("--cpu-sample-atoms", type=int, default=0)
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from one process: the library's own fan-out and reduce (ta_group)")
    ap.add_argument("--devices", default="",
                    help="--single-process: comma-separated device ids of the members (default 0..N-1)")
    ap.add_argument("--allow-peer-copy", action="store_true",
                    help="--single-process on distinct devices: accept the peer-copy reduce when RCCL fails "
                         "(default: that is an error)")
    ap.add_argument("--full-json", default="",
                    help="also write the JSON line, indented, to this file (tools/profile_bench.sh: profiles/rNN_bench_full.json)")
    ap.add_argument("--no-clock-probe", action="store_true")
    return ap.parse_args()


# Ceiling of this transform structure under the package power limit (DESIGN.md section 6,
# profiles/r04_power_clock.txt): every plan of the forward kernel runs AT the 1400 W cap
# (1352-1390 W measured), so throughput = power / energy per input byte; the most frugal plan
# measured needs 0.453 nJ/B (R0 = 12: 1389 W at 3064 GB/s), the headline's R0 = 20 0.525 nJ/B.
POWER_CAP_W = 1400.0
BEST_PLAN_NJ_PER_BYTE = 0.453


# measured (profiles/r04_energy_ubench.txt, profiles/r04_f32_slabs.txt): idle 243 W; an FP64 lane operation
# <= 33.5 pJ all-in; the HBM read not measurable (float32-slab A/B: half the bytes, same time);
# the transform needs 6.06 FP64 operations per input byte
IDLE_W, HBM_PJ_PER_BYTE, FP64_PJ_PER_OP, FP64_OPS_PER_BYTE = 243.0, 0.0, 33.5, 6.06


def model_ceiling():
    gbps = POWER_CAP_W / BEST_PLAN_NJ_PER_BYTE
    # what the transform's FP64 operations (x 0.8: the lower operating point of the frugal plans) and
    # the HBM read alone would allow: no LDS traffic, no other instruction, no waiting wave
    floor_nj = (HBM_PJ_PER_BYTE + 0.8 * FP64_OPS_PER_BYTE * FP64_PJ_PER_OP) * 1e-3
    bound = (POWER_CAP_W - IDLE_W) / floor_nj
    # cycle floors at the clock the kernel holds (profiles/r04_occupancy_counters.txt): per
    # unit and pass ~9000 cycles of vector issue per SIMD / of the LDS pipe, overlapped perfectly
    issue_floor_gbps = 160000.0 / (9000.0 / 2.16e9) * 256 / 1e9 / 2  # 10000 rows x 16 B per pair, two passes, 256 CUs
    return {"frac": gbps / HBM_PEAK_GBPS, "GBps": gbps,
            "basis": "package power cap / lowest energy per input byte measured over the forward kernel's plans "
                     "(profiles/r04_power_clock.txt); the kernel is power-bound, not issue- or bandwidth-bound",
            "issue_and_lds_floor_at_measured_clock": {"frac": issue_floor_gbps / HBM_PEAK_GBPS, "GBps": issue_floor_gbps,
                                                      "basis": "vector issue (8.1-9.1k cycles per SIMD) and LDS pipe (8.8k cycles) "
                                                               "per unit and pass, overlapped perfectly at 2.16 GHz: what a cycle "
                                                               "model allows and the power limit does not (the clock falls)"},
            "arithmetic_only": {"frac": bound / HBM_PEAK_GBPS, "GBps": bound,
                                "basis": "(cap - idle) / (the transform's FP64 operations at 0.8 x 33.5 pJ), nothing else: "
                                         "not a reachable state, the distance DESIGN.md section 6.0 item 4 describes"}}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child
    `python -m torch.distributed.run` (what the driver's own N > 1 command is) and exit with its
    status.  Called before torch is imported: this process never creates a GPU context, and the
    ranks are children, not a re-exec.  stdout is inherited, so rank 0's JSON line is this
    command's JSON line."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TA_BENCH_LAUNCHER="bench.py started torch.distributed.run itself")
    return subprocess.run(cmd, env=env).returncode


def member_rooflines(g, T, D, steps):
    """per member of a ta_group: median device time of the dominant kernel over the timed calls and the
    algorithmic HBM rate it stands for (that member's atoms x n_frames x dim x 8 bytes)"""
    out = []
    for i, (lo, hi) in enumerate(g.shards):
        if hi == lo:
            continue
        hist = g.member_context(i).timing_history(min(steps, 64))
        kms = statistics.median(m for _, m in hist)
        tms = statistics.median(t for t, _ in hist)
        b = T * (hi - lo) * D * 8
        out.append({"member": i, "device": g.devices[i], "atoms": hi - lo, "kernel_ms": kms, "whole_call_ms": tms,
                    "GBps": b / (kms * 1e-3) / 1e9, "bytes": b})
    return out


def single_process(args):
    """N GPUs from one process through ta_group: every member's column block of the ONE synthetic
    tensor resident on its device, a step = ta_group_vacf_fft (fan-out, reduce inside the library,
    (n_frames,) timeseries on the host)."""
    import numpy as np

    from transport_analysis_amd import _lib
    from oracle import synth

    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit("--devices must name --gpus devices")
    T, D = args.frames, args.dim
    a_total = args.atoms * args.gpus if args.scaling == "weak" else args.atoms
    g = _lib.Group(devices)
    distinct = len(set(devices)) == len(devices)
    if distinct and len(devices) > 1 and not args.allow_peer_copy:
        g.set_option("reduce_mode", 2)  # RCCL or an error: a curve of the wrong collective is worse than none
    g.stage_alloc_device(T, a_total, D, n_slabs=1)
    g.stage_synth(0, SEED + 3, 0, a_total * D)
    fn = {"fft": g.vacf_fft, "direct": g.vacf_direct}[args.mode]
    for _ in range(args.warmup):
        ts, _ = fn(by_particle=False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts, _ = fn(by_particle=False)
    elapsed = time.perf_counter() - t0
    members = member_rooflines(g, T, D, args.steps)
    slowest = min(members, key=lambda m: m["GBps"])
    # what the call adds to the slowest member's kernels: the reduce, the (n_frames,) copy to the host, the waits
    overhead_ms = elapsed / args.steps * 1e3 - max(m["whole_call_ms"] for m in members)
    check = {}
    if not args.no_check and args.mode == "fft":  # lag 0 of the mean = mean square of the tensor's first columns
        blk = synth.synthetic_block(SEED + 3, T, a_total * D, 0, min(a_total * D, 96))
        check["lag0_vs_numpy_block_mean_square"] = float(ts[0] / D), float(np.mean(blk * blk))
    out = {
        "metric": "VACF lag-points/sec (n_frames x n_atoms / s)", "value": T * a_total / (elapsed / args.steps),
        "unit": "lag-points/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"FFT VACF timeseries, {T} frames x {a_total} atoms x {D} float64, atoms sharded over "
                               f"{args.gpus} member(s) of ONE process (ta_group; host-facing call, result on the host)",
                   "n_frames": T, "n_atoms_total": a_total, "dim": D, "mode": args.mode,
                   "sharding": f"atoms x{args.gpus}", "member_devices": devices, "shards": g.shards,
                   "collective": {"kind": g.reduce_kind, "library": "librccl (dlopen) inside ta_group_vacf_*" if g.reduce_kind == "rccl"
                                  else "hipMemcpyPeerAsync + k_sum_partials" if g.reduce_kind == "peer-copy" else None,
                                  "ranks_seen_by_rccl": g.rccl_ranks, "members": len(devices),
                                  "fallback_note": g.reduce_note or None,
                                  "op": f"reduce(sum) of ({T},) float64 lag sums onto member 0"},
                   "library_sha16": so_sha16(), "launcher": "single process"},
        "roofline": {"bound": "hbm", "achieved": slowest["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": slowest["GBps"] / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": "k_wsplit_accum" if args.mode == "fft" else "k_band_bp_vacf",
                     "kernel_ms": slowest["kernel_ms"], "algorithmic_bytes_per_launch": slowest["bytes"],
                     "what": "the slowest member's dominant kernel (hipEvents on its own stream, median over the timed calls); "
                             "`per_member` has all of them, `reduce_and_host_ms` what the call adds to the slowest member "
                             "(the reduce, the copy of the timeseries to the host, the waits)",
                     "per_member": members, "reduce_and_host_ms": overhead_ms},
        "check": check,
        "note": "members that share a device (e.g. --devices 0,0) serialise on it",
    }
    if distinct and len(devices) > 1 and g.reduce_kind != "rccl" and not args.allow_peer_copy:
        raise SystemExit(f"the members' lag sums were not reduced by RCCL ({g.reduce_kind}; {g.reduce_note})")
    g.close()
    if not args.no_cpu_baseline and args.mode == "fft":
        out["cpu_baseline"] = cpu_baseline(args, T, D, a_total * D)
    print(json.dumps(out), flush=True)



def fft_flops(T, n_cols, M):
    """Arithmetic of the lag-sum path per launch: per column pair two M-point
    complex transforms (5 M log2 M each), the pass-B twist (6 M) and |.|^2 accumulation (3 * 2M)."""
    import math

    return ((n_cols + 1) // 2) * (2 * 5.0 * M * math.log2(M) + 12.0 * M)


def so_sha16():
    p = os.path.join(ROOT, "transport_analysis_amd", "libta_hip.so")
    try:
        return hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def recorded_traffic(key):
    """HBM bytes per launch from the PMC passes of THIS build (tools/profile_bench.sh writes the
    library's hash beside the numbers); None when the record is from another build."""
    f = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(f))
    except (OSError, ValueError):
        return None
    if rec.get("so_sha16") != so_sha16():
        return None
    e = rec.get("entries", {}).get(key)
    return e["hbm_bytes_per_launch"] if e else None


class Case:
    """One workload on one rank: slabs staged on the device, `step()` = one pass of the path."""

    def __init__(self, torch, ctx, dev, mode, T, A, D, col_offset, n_cols_total, seed, by_particle=False,
                 float32=False, helfand_fft=False, slab32=False):
        self.torch, self.ctx, self.mode, self.T, self.A, self.D = torch, ctx, mode, T, A, D
        self.helfand_fft = helfand_fft
        self.float32 = float32
        self.slab32 = slab32  # float32 device slabs under float64 arithmetic (the FFT kernels widen the rows)
        ctx.set_option("direct_f32", 1 if float32 else 0)
        ctx.set_option("stage_device_f32", 1 if (float32 or slab32) else 0)  # float32 path: float32 device slabs
        ctx.set_option("helfand_fft", 1 if helfand_fft else 0)
        self.stream = torch.cuda.current_stream().cuda_stream
        ctx.stage_alloc_device(T, A, D, n_slabs=2 if mode == "helfand" else 1)
        ctx.stage_synth(0, seed, col_offset, n_cols_total, self.stream)
        self.masses = None
        if mode == "helfand":
            # x[t] = x0 + 0.002 cumsum(v), masses cycled over O, H, H
            fm = torch.empty((T, A * D), dtype=torch.float64, device=dev)
            ctx.stage_read_dev(0, fm.data_ptr(), A * D, self.stream)
            fm = 30.0 + 0.002 * torch.cumsum(fm, dim=0)
            ctx.stage_commit_dev(1, fm.data_ptr(), A * D, 0, T, stream=self.stream)
            torch.cuda.synchronize()
            del fm
            self.masses = torch.tensor([15.999, 1.008, 1.008], dtype=torch.float64, device=dev).repeat(
                (A + 2) // 3)[:A].contiguous()
        self.lagsum = torch.zeros(T, dtype=torch.float64, device=dev)
        self.bp = torch.empty((T, A), dtype=torch.float64, device=dev) if by_particle else None

    def step(self):
        d_bp = self.bp.data_ptr() if self.bp is not None else 0
        if self.mode == "fft":
            self.ctx.vacf_fft_staged(self.lagsum.data_ptr(), d_bp, self.A, self.stream)
        elif self.mode == "direct":
            self.ctx.vacf_direct_staged(self.lagsum.data_ptr(), d_bp, self.A, self.stream)
        else:
            self.ctx.helfand_msd_staged(self.masses.data_ptr(), 1.0, self.lagsum.data_ptr(), d_bp, self.A,
                                        self.stream)

    def on_short_kernel(self):
        """up to 64 frames (float64 arithmetic on float64 slabs): k_short (short_kernels.hpp) -- every by-particle array, the
        O(T^2) lag sums, and the FFT path's lag sums up to 48 frames"""
        if self.T > 64 or self.float32 or self.slab32 or self.helfand_fft:
            return False
        return self.bp is not None or self.mode != "fft" or self.T <= 48

    def on_mid_kernel(self):
        """65 < n_frames <= 512, float64 on float64 slabs: k_mid (mid_kernels.hpp) runs the windowed VACF from 97 frames and
        the Einstein-Helfand sums from 97 to 128 ("mid_max")"""
        if self.float32 or self.slab32 or self.helfand_fft or self.mode == "fft":
            return False
        return 97 <= self.T <= (512 if self.mode == "direct" else 128)

    def on_matrix_cores(self):
        """lag sums alone of the O(T^2) correlators: FP64 MFMA band kernel (band_kernels.hpp); the float32 option's
        Helfand lag sums: FP32 MFMA (band32_kernels.hpp)"""
        if self.mode == "helfand" and self.helfand_fft:
            return False
        if self.on_short_kernel() or self.on_mid_kernel():
            return False
        # ("direct_mfma" 1: the vector kernel below these lengths, api.hip direct_impl / profiles/r06_direct_mid_sweep.txt)
        if self.T < (513 if self.mode == "direct" else 448 if self.float32 else 352):
            return False
        if self.bp is not None:  # by-particle arrays: the float32 Helfand form (dim = 3) and both float64 forms
            return self.mode == "helfand" or not self.float32
        return self.mode == "helfand" or (self.mode == "direct" and not self.float32)

    def kernel_name(self):
        if self.on_short_kernel():
            return "k_short"
        if self.on_mid_kernel():
            return "k_mid"
        if self.on_matrix_cores():
            if self.bp is not None:
                return "k_band32_tp" if self.float32 else ("k_band_bp_vacf" if self.mode == "direct" else "k_band_bp_helf")
            if self.mode == "helfand":
                return "k_band32_tp" if self.float32 else "k_band_bp_helf"
            return "k_band_bp_vacf"
        if self.mode != "fft" and not (self.mode == "helfand" and self.helfand_fft):
            return "k_direct"
        if self.T > 163840:
            return "k_direct"
        if self.T <= 512:
            return "k_w1_bp" if self.bp is not None else "k_w1_accum"
        return "k_wsplit_accum+k_winverse" if self.bp is not None else "k_wsplit_accum"


def timed(torch, dist, world, steps, warmup, fn):
    """W warm-ups, then exactly K steps between two fences; wall seconds (max over ranks)."""
    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        fn()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def kernel_split(ctx, case, torch):
    """[{name, ms}] of an extra (untimed) step, from the library's event timeline."""
    ctx.set_option("timeline", 1)
    try:
        for _ in range(3):  # back to back like the timed steps; the last one is read
            case.step()
        torch.cuda.synchronize()
        return [{"name": n, "ms": round(ms, 4)} for n, ms in ctx.kernel_timeline() if n != "end"]
    finally:
        ctx.set_option("timeline", 0)


def roofline_of(case, kernel_ms, helfand_fft=False, float32=False):
    """kernel_ms: device time the bytes / flops are divided by -- the dominant kernel's for a
    path that IS one kernel (FFT lag sums, direct correlators), the whole call's otherwise."""
    T, A, D = case.T, case.A, case.D
    from transport_analysis_amd import _lib

    bytes_algo = T * A * D * (4 if getattr(case, "slab32", False) else 8) * (2 if case.mode == "helfand" else 1)
    if case.bp is not None:
        bytes_algo += T * A * 8  # the by-particle array written once
    gbps = bytes_algo / (kernel_ms * 1e-3) / 1e9
    if case.on_short_kernel():
        # k_short (short_kernels.hpp): O(T^2) in registers, HBM-bound: T A D 8 bytes in (x 2 with positions), T A 8 out
        fl = (3.0 * D * A * T * (T - 1) / 2) if case.mode == "helfand" else (2.0 * D * A * T * (T + 1) / 2)
        tf = fl / (kernel_ms * 1e-3) / 1e12
        return {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                "traffic": None, "kernel": "k_short", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_algo,
                "valu": {"achieved_tflops": tf, "peak": FP64_PEAK_TFLOPS, "frac": tf / FP64_PEAK_TFLOPS,
                         "algorithmic_flops_per_launch": fl}}
    if case.mode == "fft" or helfand_fft:
        M = _lib.fft_plan_info(T)["M"]
        fl = fft_flops(T, A * D, M)
        tf = fl / (kernel_ms * 1e-3) / 1e12
        name = case.kernel_name()
        return {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": gbps / HBM_PEAK_GBPS, "traffic": None, "kernel": name, "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": bytes_algo,
                "valu": {"achieved_tflops": tf, "peak": FP64_PEAK_TFLOPS, "frac": tf / FP64_PEAK_TFLOPS,
                         "algorithmic_flops_per_launch": fl}}
    # direct correlators: vector kernels bound by the FP issue rate, matrix-core kernels by the MFMA pipe:
    # windowed VACF 2*D*A*T(T+1)/2 flop, Helfand 3*D*A*T(T-1)/2; HBM is touched once
    fl = (2.0 * D * A * T * (T + 1) / 2) if case.mode == "direct" else (3.0 * D * A * T * (T - 1) / 2)
    peak = FP32_PEAK_TFLOPS if float32 else FP64_PEAK_TFLOPS
    tf = fl / (kernel_ms * 1e-3) / 1e12
    if case.on_matrix_cores():
        # v_mfma_f64_16x16x4_f64: 78.6 TFLOP/s dense at 2.4 GHz (77.3 measured back to back,
        # profiles/r04_mfma_f64_ubench.txt); v_mfma_f32_16x16x4_f32: 157.3.  The flop counted for the windowed
        # VACF are the useful ones (T(T+1)/2 lag products per column)
        mpeak = FP32_MFMA_PEAK_TFLOPS if float32 else FP64_MFMA_PEAK_TFLOPS
        out = {"bound": "mfma", "achieved": tf, "peak": mpeak, "unit": "TFLOP/s",
               "frac": tf / mpeak, "traffic": None, "kernel": case.kernel_name(), "kernel_ms": kernel_ms,
               "algorithmic_flops_per_launch": fl, "hbm_GBps_for_reference": gbps}
        if case.mode == "helfand":
            # `achieved` / `frac` = what the matrix pipe ISSUES: 2 flop per term (the time-packed kernels k_band_bp_helf /
            # k_band32_tp use all four k-slots for products; the norms are vector adds).  The reference's own arithmetic
            # (difference, square, add = 3 flop per term; what the vector kernel's line counts) is
            # `reference_flops_*`: a larger number that says nothing about the pipe
            slots = 1.0
            issued = 2.0 * slots * D * A * T * (T - 1) / 2 / (kernel_ms * 1e-3) / 1e12
            out.update({"achieved": issued, "frac": issued / mpeak,
                        "issued_flops_per_launch": 2.0 * slots * D * A * T * (T - 1) / 2,
                        "reference_flops_tflops": tf, "reference_flops_frac": tf / mpeak})
        return out
    return {"bound": "valu", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
            "traffic": None, "kernel": case.kernel_name(), "kernel_ms": kernel_ms,
            "algorithmic_flops_per_launch": fl, "hbm_GBps_for_reference": gbps}


def cpu_baseline(args, T, D, n_cols_total, seconds=12.0):
    """NumPy oracle (reference control flow: per-atom loop, tidynamics-style FFT) on ONE core
    over the first atoms of the SAME synthetic tensor; linear in the atom count."""
    from oracle import numpy_oracle as orc
    from oracle import synth

    all_cpus = None
    try:
        all_cpus = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {sorted(all_cpus)[0]})
    except Exception:
        pass
    a = args.cpu_sample_atoms or max(8, int(8.0e7 * seconds / 12.0 // T))  # ~10-15 s of CPU work at the default
    v = synth.synthetic_block(SEED + 3, T, n_cols_total, 0, a * D).reshape(T, a, D)
    t0 = time.perf_counter()
    _, port_ts = orc.vacf_fft(v)
    dt = time.perf_counter() - t0
    what = (f"oracle.numpy_oracle.vacf_fft on {T} frames x the first {a} atoms x {D} of the benchmark tensor "
            f"(oracle.synth, seed {SEED + 3})")
    out = {"value": T * a / dt, "unit": "lag-points/s", "cores": 1, "kind": "port", "sample": what,
           "seconds": round(dt, 2),
           "note": "the reference itself cannot be timed here: it imports MDAnalysis and tidynamics, neither of "
                   "which is installed on any box of this pool; the port restates its _conclude_fft (per-atom "
                   "loop over tidynamics-style FFT autocorrelations) and is pinned by the reference's fixtures"}
    # second line, the "best CPU" figure: the library's own opt-in CPU backend
    # (csrc/cpu_backend.cpp, C++/OpenMP, behind the same C symbols) on every usable host core, on its own block of the
    # same tensor (generated by ta_stage_synth on the host), cross-checked against the one-core port's lag sums
    try:
        import numpy as np

        from oracle import parallel
        from transport_analysis_amd import _lib

        if all_cpus:
            os.sched_setaffinity(0, all_cpus)
        n = parallel.usable_cpus()
        c = _lib.Context("cpu")
        try:
            c.set_option("cpu_threads", n)
            c.stage_alloc(T, a, D, dtype=np.float64)
            c.stage_synth(0, SEED + 3, 0, n_cols_total)
            t0 = time.perf_counter()
            ts_small, _ = c.vacf_fft(by_particle=False)
            dt_small = time.perf_counter() - t0
            agree = float(np.max(np.abs(ts_small - port_ts)) / np.max(np.abs(port_ts)))
            # ~the one-core sample's duration of work for the whole team, within 6 GB of host slab
            a_all = int(min(n_cols_total // D, 6e9 // (T * D * 8), max(a, a * (dt / max(dt_small, 1e-3)) * 0.8)))
            c.stage_alloc(T, a_all, D, dtype=np.float64)
            c.stage_synth(0, SEED + 3, 0, n_cols_total)
            t0 = time.perf_counter()
            c.vacf_fft(by_particle=False)
            dtp = time.perf_counter() - t0
        finally:
            c.close()
        out["all_cores"] = {"value": T * a_all / dtp, "unit": "lag-points/s", "cores": n, "kind": "port",
                            "impl": "libta_hip.so's opt-in CPU backend (csrc/cpu_backend.cpp: C++/OpenMP, own radix-4 transform), "
                                    "ta_ctx_create(TA_DEVICE_CPU)",
                            "sample": f"the first {a_all} atoms", "seconds": round(dtp, 2),
                            "agrees_with_one_core_port": agree}
    except Exception as e:  # optional second line
        out["all_cores"] = {"error": str(e)[:200]}
    return out


def host_path(torch, _lib, dev_index, T, D):
    """PCIe-inclusive rate of the drop-in path: float32 pinned slab -> ta_stage_commit (H2D in
    64 MiB pieces + on-device transposition) -> ta_vacf_fft, on an atom block.  Never `value`."""
    import numpy as np

    A = 20000
    ctx = _lib.Context(dev_index)
    try:
        (slab,) = ctx.stage_alloc(T, A, D, n_slabs=1, dtype=np.float32)
        rng = np.random.default_rng(5)
        blk = rng.standard_normal((256, A, D), dtype=np.float32)
        for t in range(0, T, 256):
            slab[t:t + 256] = blk[: min(256, T - t)]
        out = {}
        for rep in range(2):
            t0 = time.perf_counter()
            ctx.stage_commit(0, T)
            ts, _ = ctx.vacf_fft(by_particle=False)
            dt = time.perf_counter() - t0
        out = {"value": T * A / dt, "unit": "lag-points/s", "seconds": dt,
               "what": f"float32 pinned slab {T}x{A}x{D} -> commit (PCIe + transpose) -> ta_vacf_fft, second of two runs",
               "pcie_GBps": T * A * D * 4 / dt / 1e9}
        return out
    finally:
        ctx.close()


def staging_cost(torch, ctx, dev, T, A, D):
    """What the timed region leaves out on the device side: (a) k_relayout, the transposition of
    committed frame-major frames into the pair-major slab (ta_stage_commit_dev on the whole
    tensor); (b) a complete ta_vacf_fft_dev call on a frame-major device tensor (transposition
    into the context's scratch slab + the FFT path)."""
    fm = torch.empty((T, A * D), dtype=torch.float64, device=dev)
    fm.normal_()
    stream = torch.cuda.current_stream().cuda_stream
    ctx.stage_alloc_device(T, A, D, n_slabs=1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    for _ in range(3):
        ev[0].record()
        ctx.stage_commit_dev(0, fm.data_ptr(), A * D, 0, T, stream=stream)
        ev[1].record()
        torch.cuda.synchronize()
        ms.append(ev[0].elapsed_time(ev[1]))
    relayout_ms = min(ms)
    ctx.stage_free()
    lag = torch.zeros(T, dtype=torch.float64, device=dev)
    ctx.set_option("timeline", 1)
    try:
        tot = []
        for _ in range(3):
            ctx.vacf_fft_dev(fm.data_ptr(), T, A, D, A * D, lag.data_ptr(), 0, 0, stream)
            torch.cuda.synchronize()
            tot.append(ctx.last_timing()[0])
        split = [{"name": n, "ms": round(m, 4)} for n, m in ctx.kernel_timeline() if n != "end"]
    finally:
        ctx.set_option("timeline", 0)
    nbytes = T * A * D * 8
    # (c) the same tensor as float32 frame-major rows (MDAnalysis' dtype; what a caller that decodes
    # on the GPU holds): ta_stage_commit_dev into a float32 device slab + ta_vacf_fft_staged, the
    # FFT kernels widening the rows themselves
    f32 = {}
    try:
        fm32 = fm.to(torch.float32)
        del fm
        ctx.set_option("stage_device_f32", 1)
        ctx.stage_alloc_device(T, A, D, n_slabs=1)
        t32 = []
        for _ in range(3):
            ev[0].record()
            ctx.stage_commit_dev(0, fm32.data_ptr(), A * D, 0, T, dtype="float32", stream=stream)
            ctx.vacf_fft_staged(lag.data_ptr(), 0, 0, stream)
            ev[1].record()
            torch.cuda.synchronize()
            t32.append(ev[0].elapsed_time(ev[1]))
        f32 = {"vacf_fft_from_float32_frame_major_ms": min(t32),
               "vacf_fft_from_float32_frame_major_note": "ta_stage_commit_dev(TA_F32) into a float32 device slab "
               "(\"stage_device_f32\") + ta_vacf_fft_staged; float64 arithmetic"}
        del fm32
    except Exception as e:  # never lose the float64 figures over the extra line
        f32 = {"vacf_fft_from_float32_frame_major_error": str(e)[:200]}
    finally:
        ctx.set_option("stage_device_f32", 0)
        ctx.stage_free()
    del lag
    ctx.trim()
    torch.cuda.empty_cache()
    return {"tensor": f"{T} x {A} x {D} float64 frame-major on the device ({nbytes / 1e9:.1f} GB)", **f32,
            "k_relayout_ms": relayout_ms,
            "k_relayout_GBps": 2 * nbytes / (relayout_ms * 1e-3) / 1e9,  # read + write
            "vacf_fft_dev_ms": min(tot), "vacf_fft_dev_kernels": split,
            "vacf_fft_dev_lag_points_per_s": T * A / (min(tot) * 1e-3)}


def _lib_threads():
    from transport_analysis_amd import _lib

    return int(_lib.lib().ta_stage_threads())


def host_path_by_particle(dev_index, T, D):
    """The drop-in CLASS end to end, the reference's default output included:
    VelocityAutocorr(ArrayUniverse(...), fft=True).run() at T x 50000 x D, float32 frames through
    pinned staging, results.vacf_by_particle in pinned memory.  The frame loop is the stand-in
    AnalysisBase's Python loop (with MDAnalysis: the trajectory reader); `conclude_s` is what
    the library adds after the last frame: final commit, compute, device->host."""
    import numpy as np

    from transport_analysis_amd import VelocityAutocorr
    from transport_analysis_amd._mini_mda import ArrayUniverse

    A = 50000
    rng = np.random.default_rng(11)
    blk = rng.standard_normal((250, A, 3), dtype=np.float32)
    vel = np.empty((T, A, 3), dtype=np.float32)
    for t in range(0, T, 250):
        vel[t:t + 250] = blk[: min(250, T - t)]
    u = ArrayUniverse(velocities=vel, positions=None)
    res = {}
    for rep in range(2):
        an = VelocityAutocorr(u.atoms, fft=True, device=dev_index)
        marks = {}
        orig_conclude = an._conclude

        def timed_conclude():
            marks["loop_end"] = time.perf_counter()
            orig_conclude()
            marks["end"] = time.perf_counter()

        an._conclude = timed_conclude
        orig_prepare = an._prepare

        def timed_prepare():
            a0 = time.perf_counter()
            orig_prepare()
            marks["prepare_s"] = time.perf_counter() - a0

        an._prepare = timed_prepare
        t0 = time.perf_counter()
        an.run()
        loop_s, conclude_s = marks["loop_end"] - t0, marks["end"] - marks["loop_end"]
        bp = an.results.vacf_by_particle
        res = {"what": f"VelocityAutocorr(fft=True).run() through the class, {T} x {A} x {D} float32 frames, "
                       f"vacf_by_particle {bp.shape[0]} x {bp.shape[1]} float64 in pinned memory; second of two runs",
               "frame_loop_s": loop_s, "prepare_s": marks["prepare_s"], "frames_s": loop_s - marks["prepare_s"],
               "frame_loop_note": "frame_loop_s = run() up to _conclude = prepare_s (pinned slab of the input's size page-locked "
                                  "and zeroed, device slab) + frames_s (the per-frame fills: ta_stage_frame on "
                                  f"{_lib_threads()} host threads; commits are queued to a worker thread)",
               "conclude_s": conclude_s, "total_s": loop_s + conclude_s,
               "bytes_in": T * A * D * 4, "bytes_out": bp.nbytes,
               "pcie_floor_s_at_48GBps": (T * A * D * 4 + bp.nbytes) / 48e9,
               "out_GBps_if_conclude_were_all_copy": bp.nbytes / conclude_s / 1e9,
               "value": T * A / (loop_s + conclude_s), "unit": "lag-points/s"}
        del an, bp
    return res


def cpu_rehearsal(args, world, rank):
    """TA_BENCH_CPU=1: the N-rank path of this file with every rank on the library's opt-in CPU backend
    (ta_ctx_create(TA_DEVICE_CPU)) and gloo -- launcher, WORLD_SIZE checks, atom sharding under both scalings, the
    all-reduce of the lag sums, max-over-ranks timing, ONE JSON line from rank 0 -- at rank counts a one-GPU box may
    not put on its card (the driver's N = 8).  A REHEARSAL OF THE PLUMBING: its `value` is host arithmetic and is
    labelled so; it is never the benchmark."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from transport_analysis_amd import _lib
    from transport_analysis_amd.dist import atom_shard, reduce_lagsum

    if args.mode != "fft" or args.by_particle:
        raise SystemExit("TA_BENCH_CPU=1 rehearses --mode fft lag sums")
    grouped = world > 1
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    T, D = args.frames, args.dim
    if args.scaling == "weak":
        a_total = args.atoms * world
        lo, hi = args.atoms * rank, args.atoms * (rank + 1)
    else:
        a_total = args.atoms
        lo, hi = atom_shard(a_total, rank, world)
    A = hi - lo
    ctx = _lib.Context("cpu")
    ctx.set_option("cpu_threads", max(1, (os.cpu_count() or 1) // max(world, 1)))
    if A:
        ctx.stage_alloc(T, A, D, dtype=np.float64)
        ctx.stage_synth(0, SEED + 3, lo * D, a_total * D)
    result = {}

    def step():
        lag = np.zeros(T)
        if A:
            ts, _ = ctx.vacf_fft(by_particle=False)
            lag = ts * A  # this rank's lag SUMS
        result["ts"] = reduce_lagsum(torch.from_numpy(lag), a_total)

    for _ in range(args.warmup):
        step()
    if grouped:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [{"rank": rank, "atoms": A, "atom_range": [lo, hi]}]
    collective = None
    if grouped:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
        collective = {"backend": dist.get_backend(), "library": "gloo (CPU-backend rehearsal, lag sums on the host)",
                      "ranks": dist.get_world_size(), "op": f"all_reduce(sum) of ({T},) float64 lag sums on the host",
                      "launcher": os.environ.get("TA_BENCH_LAUNCHER", "torch.distributed.run (started by the caller)")}
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # the reduced series against the whole tensor computed by this rank alone (small rehearsal shapes only)
    check = None
    if T * a_total * D <= 4_000_000:
        ctx.stage_alloc(T, a_total, D, dtype=np.float64)
        ctx.stage_synth(0, SEED + 3, 0, a_total * D)
        whole, _ = ctx.vacf_fft(by_particle=False)
        check = float(np.max(np.abs(result["ts"].numpy() - whole)) / np.max(np.abs(whole)))
    ctx.close()
    print(json.dumps({
        "metric": "VACF lag-points/sec (n_frames x n_atoms / s)", "value": T * a_total / (elapsed / args.steps),
        "unit": "lag-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "rehearsal": "TA_BENCH_CPU=1: every rank on the library's CPU backend; plumbing only, NOT a GPU measurement",
        "config": {"workload": f"CPU-backend rehearsal of the {world}-rank path: FFT VACF timeseries, {T} frames x {a_total} atoms x {D}",
                   "n_frames": T, "n_atoms_total": a_total, "n_atoms_this_rank": A, "dim": D, "mode": "fft", "by_particle": False,
                   "sharding": f"atoms x{world}", "collective": collective, "per_rank": per_rank},
        "roofline": None, "cpu_baseline": None,
        "check": {"reduced_series_vs_one_rank_scale_rel": check}}), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.single_process:
        # no launcher around us: be the launcher (a child process; nothing here has touched a GPU)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("TA_BENCH_CPU") == "1" and not args.single_process:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        return cpu_rehearsal(args, world, rank)
    if args.single_process:
        if world > 1:
            raise SystemExit("--single-process drives the GPUs itself: do not launch it under torchrun")
        return single_process(args)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # TA_BENCH_ONE_GPU=1: rehearsal of the N > 1 path on a one-GPU box (tests/test_gpu_dist.py):
    # every rank on device 0, gloo instead of RCCL, lag sums reduced through the host
    one_gpu = os.environ.get("TA_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # a process group for N > 1; for N = 1 only on request (TA_BENCH_FORCE_DIST=1 under torchrun:
    # the RCCL branch of the step on a one-GPU box)
    grouped = world > 1 or (os.environ.get("TA_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    # which GPU is this rank on?  (uuid where torch has it, else the PCI address)
    props = torch.cuda.get_device_properties(dev)
    dev_id = str(getattr(props, "uuid", "")) or ""
    if not dev_id or set(dev_id) <= set("0-"):
        dev_id = "pci:%s:%s:%s" % tuple(getattr(props, a, "?") for a in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    rank_devices = [f"{local_rank}|{dev_id}"]
    if grouped:
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_devices[0])
        rank_devices = gathered
        if not one_gpu and len(set(d.split("|", 1)[1] for d in rank_devices)) != world:
            raise SystemExit(f"ranks share a GPU: {rank_devices}")

    from transport_analysis_amd import _lib
    from transport_analysis_amd.dist import atom_shard, reduce_lagsum

    T, D = args.frames, args.dim
    if args.float32 and args.mode == "fft":
        raise SystemExit("--float32 applies to --mode direct / helfand")
    if args.helfand_fft and (args.mode != "helfand" or args.float32):
        raise SystemExit("--helfand-fft applies to --mode helfand without --float32")
    if args.scaling == "weak":
        a_total = args.atoms * world
        lo, hi = args.atoms * rank, args.atoms * (rank + 1)
    else:
        a_total = args.atoms
        lo, hi = atom_shard(a_total, rank, world)
    A = hi - lo
    seed = SEED + {"fft": 3, "direct": 4, "helfand": 5}[args.mode]
    ctx = _lib.Context(local_rank)
    if args.slab32 and args.mode != "fft":
        raise SystemExit("--slab32 applies to --mode fft")
    case = Case(torch, ctx, dev, args.mode, T, A, D, lo * D, a_total * D, seed, args.by_particle,
                args.float32, args.helfand_fft, args.slab32)

    result = {}
    red_events, red_host = [], []

    def step():
        case.step()
        if one_gpu and world > 1:
            t0 = time.perf_counter()
            result["ts"] = reduce_lagsum(case.lagsum.cpu(), a_total)
            red_host.append((time.perf_counter() - t0) * 1e6)
        elif grouped:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            result["ts"] = reduce_lagsum(case.lagsum, a_total)  # device tensor in and out: RCCL
            e1.record()
            red_events.append((e0, e1))
        else:
            result["ts"] = reduce_lagsum(case.lagsum, a_total)

    elapsed = timed(torch, dist, world if grouped else 1, args.steps, args.warmup, step)
    reduce_us = None
    if red_events:
        torch.cuda.synchronize()
        reduce_us = statistics.median(a.elapsed_time(b) * 1e3 for a, b in red_events[-args.steps:])
    elif red_host:
        reduce_us = statistics.median(red_host[-args.steps:])
    hist = ctx.timing_history(min(args.steps, 64))
    kernel_ms = statistics.median(m for _, m in hist)
    total_ms = statistics.median(t for t, _ in hist)
    # the clock the headline kernel holds: >= 2 s of back-to-back launches of its stamped build
    clock = None
    if args.mode == "fft" and not args.by_particle and not args.no_clock_probe and world == 1:
        try:
            clock = ctx.clock_probe(max(20, int(2500.0 / max(kernel_ms, 0.05))))
        except Exception as e:  # plans without a stamped build
            clock = {"error": str(e)[:160]}

    # validity of the timed result (N = 1): the staged tensor IS the NumPy generator's (bit for
    # bit, on a block), and a few lags agree with plain torch reductions over the whole tensor
    check = {}
    if world == 1 and not args.no_check and args.mode in ("fft", "direct"):
        from oracle import synth

        fm = torch.empty((T, A * D), dtype=torch.float64, device=dev)
        ctx.stage_read_dev(0, fm.data_ptr(), A * D, case.stream)
        torch.cuda.synchronize()
        blk = synth.synthetic_block(seed, min(T, 64), a_total * D, lo * D, lo * D + min(A * D, 96))
        if args.slab32:  # the slab holds the tensor rounded once to float32
            blk = blk.astype(np.float32).astype(np.float64)
        check["generator_bit_exact_vs_numpy"] = bool(
            np.array_equal(fm[: blk.shape[0], : blk.shape[1]].cpu().numpy(), blk))
        scale = float((fm * fm).sum().item()) / T
        errs = []
        for k in (0, 1, T // 2, T - 1):
            ref = float((fm[: T - k] * fm[k:]).sum().item()) / (T - k)
            errs.append(abs(float(case.lagsum[k].item()) - ref) / scale)
        check["max_scale_rel_err_vs_torch_lags"] = max(errs)
        del fm

    # every rank's dominant-kernel time (its own hipEvents) and bytes, then the group is done: what
    # follows on rank 0 (CPU baseline, JSON) needs no collective and must not keep the others waiting
    per_rank = [{"rank": rank, "atoms": A, "kernel_ms": kernel_ms, "whole_call_ms": total_ms, "reduce_us": reduce_us}]
    collective = None
    if grouped:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
        collective = {"backend": dist.get_backend(), "library": "RCCL (torch.distributed backend nccl)" if dist.get_backend() == "nccl"
                      else "gloo (one-GPU rehearsal, lag sums through the host)", "ranks": dist.get_world_size(),
                      "op": f"all_reduce(sum) of ({T},) float64 lag sums, device tensor in and out" if not one_gpu
                      else f"all_reduce(sum) of ({T},) float64 lag sums on the host",
                      "launcher": os.environ.get("TA_BENCH_LAUNCHER", "torch.distributed.run (started by the caller)")}
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    composite = args.by_particle or args.helfand_fft
    roof = roofline_of(case, total_ms if composite else kernel_ms, args.helfand_fft, args.float32)
    if composite:
        roof["kernel"] = "whole call"
    key = (f"{args.mode}_{T}x{A}x{D}" + ("_bp" if args.by_particle else "") + ("_hfft" if args.helfand_fft else "")
           + ("_f32" if args.float32 else "") + ("_slab32" if args.slab32 else ""))
    roof["traffic"] = recorded_traffic(key)
    used_ms = total_ms if composite else kernel_ms
    roof["frac_kernel"], roof["frac_call"] = roof["frac"] * used_ms / kernel_ms, roof["frac"] * used_ms / total_ms
    if world > 1:
        # N > 1: `achieved` is the SLOWEST rank's rate (its own bytes over its own kernel time); the reduce is apart
        rates = []
        for r in per_rank:
            b = roof["algorithmic_bytes_per_launch"] * r["atoms"] / A if "algorithmic_bytes_per_launch" in roof else None
            ms_r = r["whole_call_ms"] if composite else r["kernel_ms"]
            rates.append({"rank": r["rank"], "atoms": r["atoms"], "kernel_ms": ms_r,
                          "GBps": (b / (ms_r * 1e-3) / 1e9) if b else None, "reduce_us": r["reduce_us"]})
        roof["per_rank"] = rates
        if all(x["GBps"] for x in rates):
            slow = min(rates, key=lambda x: x["GBps"])
            roof.update({"achieved": slow["GBps"], "frac": slow["GBps"] / HBM_PEAK_GBPS, "kernel_ms": slow["kernel_ms"],
                         "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"] * slow["atoms"] / A,
                         "what": "slowest rank's dominant kernel, hipEvents on its launch stream; the all-reduce is `reduce_us`"})
        roof["reduce_us_median_over_ranks"] = (statistics.median(r["reduce_us"] for r in per_rank)
                                               if all(r["reduce_us"] is not None for r in per_rank) else None)
    if not args.no_kernel_split:
        try:
            roof["kernels"] = kernel_split(ctx, case, torch)
        except Exception as e:
            roof["kernels"] = {"error": str(e)[:200]}
    workload = ({"fft": "FFT VACF", "direct": "windowed (direct) VACF",
                 "helfand": "Helfand MSD (FFT option)" if args.helfand_fft else "Helfand MSD"}[args.mode]
                + (" with the by-particle array" if args.by_particle else " timeseries")
                + (f", {T} frames x {args.atoms} atoms x {D} float64 per GPU" if args.scaling == "weak"
                   else f", {T} frames x {a_total} atoms x {D} float64 in total, atoms sharded over {world} GPU(s)")
                + (" (BASELINE configs[2] tensor on one GPU)" if (T, a_total, D, args.mode, world) == (10000, 100000, 3, "fft", 1)
                   else " (BASELINE configs[2])" if (T, a_total, D, args.mode, args.scaling) == (10000, 100000, 3, "fft", "strong")
                   else ""))
    out = {
        "metric": "VACF lag-points/sec (n_frames x n_atoms / s)",
        "value": T * a_total / (elapsed / args.steps),
        "unit": "lag-points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32 (f64 inputs and accumulators)" if args.float32 else
                 "f64 (float32 device slab, widened exactly)" if args.slab32 else "f64",
        "data": "synthetic",
        "config": {
            "workload": workload, "n_frames": T, "n_atoms_total": a_total, "n_atoms_this_rank": A, "dim": D,
            "mode": args.mode, "by_particle": bool(args.by_particle), "sharding": f"atoms x{world}",
            "fft_plan": _lib.fft_plan_info(T), "input": "library device slab (pair-major), ta_stage_synth: zero-mean unit-variance Irwin-Hall(8) "
                                                        "variates of splitmix64 fields (integer sums: bit-identical "
                                                        "NumPy twin oracle/synth.py), not SURVEY 8(d)'s Box-Muller normal",
            "library_sha16": so_sha16(),
            "rank_devices": rank_devices,
            "collective": collective,
        },
        "roofline": roof,
        "device_ms": {"whole_call_median": total_ms, "dominant_kernel_median": kernel_ms},
        "check": check,
    }
    if clock is not None:
        out["device_ms"]["effective_mhz"] = clock.get("mhz")
        out["device_ms"]["clock_probe"] = clock
    if args.mode == "fft" and not composite:
        out["roofline"]["model_ceiling"] = model_ceiling()
    if reduce_us is not None:
        out["reduce_us"] = reduce_us  # the all-reduce of the (n_frames,) lag sums alone, median per step
    if world == 1 and not args.no_other_configs and args.mode == "fft" and not args.by_particle:
        out["other_configs"] = other_configs(torch, dist, _lib, ctx, dev)
    if world == 1 and not args.no_host_path and args.mode == "fft":
        try:
            del case
            ctx.stage_free()
            ctx.trim()
            torch.cuda.empty_cache()
            out["staging"] = staging_cost(torch, ctx, dev, T, A, D)
        except Exception as e:
            out["staging"] = {"error": str(e)[:200]}
        try:
            ctx.stage_free()
            ctx.trim()
            torch.cuda.empty_cache()
            out["host_path"] = host_path(torch, _lib, local_rank, T, D)
        except Exception as e:
            out["host_path"] = {"error": str(e)[:200]}
        try:
            out["host_path_by_particle"] = host_path_by_particle(local_rank, T, D)
        except Exception as e:
            out["host_path_by_particle"] = {"error": str(e)[:200]}
    if not args.no_cpu_baseline and args.mode == "fft":
        # N > 1: the other ranks are done; a shorter sample (the run is N times the data already)
        out["cpu_baseline"] = cpu_baseline(args, T, D, a_total * D, seconds=12.0 if world == 1 else 4.0)
    if "other_configs" in out:
        # LAST key of the line (a log that keeps only the tail of stdout still has every config's figures)
        # per row: `frac_kernel` = the path's algorithmic bytes (flops) over its dominant kernel alone, `frac_call` = over the
        # whole call (every launch of the path); `frac` is the one the row's roofline object is quoted with
        head = {"workload": "headline: " + workload, "ms": round(ms_per_step, 4), "frac": round(roof["frac"], 4),
                "frac_kernel": round(roof["frac_kernel"], 4), "frac_call": round(roof["frac_call"], 4), "bound": roof["bound"],
                "input": "the library's pair-major device slab"}
        if isinstance(out.get("staging"), dict) and "vacf_fft_dev_ms" in out["staging"]:
            head["ms_from_frame_major_input"] = round(out["staging"]["vacf_fft_dev_ms"], 4)  # the reference's (T, A, D) layout on the device: + k_relayout
        out["summary"] = [head] + [
            {"workload": c["workload"], "ms": round(c["ms_per_step"], 4), "frac": round(c["roofline"]["frac"], 4),
             "frac_kernel": round(c["roofline"]["frac_kernel"], 4), "frac_call": round(c["roofline"]["frac_call"], 4),
             "bound": c["roofline"]["bound"]} if "error" not in c else {"workload": c["workload"], "error": c["error"][:80]}
            for c in out["other_configs"]]
    line = json.dumps(out)
    print(line, flush=True)
    if args.full_json:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.full_json)), exist_ok=True)
            with open(args.full_json, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            print(f"--full-json: {e}", file=sys.stderr)


def other_configs(torch, dist, _lib, ctx, dev):
    """Short runs of the other BASELINE configs on this GPU (rank 0, N = 1): configs[1], configs[2]
    with the by-particle array (the reference's default output), configs[3], and one GPU's
    share of configs[4] (20000 frames x 25000 atoms, float32 path)."""
    res = []
    specs = [
        ("configs[1]: FFT VACF timeseries 1000 x 10000 x 3", "fft", 1000, 10000, False, False, False, 10, 3),
        ("configs[2] one-eighth share, 10000 x 12500 x 3 (what each of 8 GPUs computes under --scaling strong; measured on "
         "ONE GPU: a prediction of the per-GPU term of that curve, the all-reduce not included)", "fft", 10000, 12500, False, False, False, 10, 3),
        ("configs[2] shape with vacf_by_particle: FFT VACF 10000 x 100000 x 3", "fft", 10000, 100000, True, False, False, 3, 1),
        ("configs[3]: windowed (direct) VACF 5000 x 50000 x 3", "direct", 5000, 50000, False, False, False, 3, 1),
        ("configs[3] shape with vacf_by_particle (the class default output; FP64 matrix cores, k-slots from the time axis): windowed VACF 5000 x 50000 x 3", "direct", 5000, 50000, True, False, False, 2, 1),
        ("configs[4] per-GPU share: Helfand MSD 20000 x 25000 x 3, float64 (FP64 matrix cores, k-slots from the time axis)", "helfand", 20000, 25000, False, False, False, 2, 1),
        ("configs[4] per-GPU share, float64 with visc_by_particle (the class default output; FP64 matrix cores): 20000 x 25000 x 3", "helfand", 20000, 25000, True, False, False, 2, 1),
        ("configs[4] per-GPU share: Helfand MSD 20000 x 25000 x 3, float32 path (FP32 matrix cores, k-slots from the time axis)", "helfand", 20000, 25000, False, True, False, 2, 1),
        ("configs[4] per-GPU share, float32 path with visc_by_particle (the class default output; FP32 matrix cores): 20000 x 25000 x 3", "helfand", 20000, 25000, True, True, False, 2, 1),
        ("configs[4] per-GPU share, helfand_fft option (float64): 20000 x 25000 x 3", "helfand", 20000, 25000, False, False, True, 3, 1),
        ("long trajectory: FFT VACF timeseries 20000 x 25000 x 3", "fft", 20000, 25000, False, False, False, 5, 1),
        ("long trajectory with vacf_by_particle: FFT VACF 20000 x 25000 x 3", "fft", 20000, 25000, True, False, False, 3, 1),
        # MDAnalysis data is float32 at the source: the same tensor rounded once to float32, kept as
        # float32 in the device slab (12 GB), float64 arithmetic; `frac` is priced with the 12 B per
        # lag-point this input has
        ("float32 device slab (MDAnalysis' dtype), float64 arithmetic: FFT VACF timeseries 10000 x 100000 x 3", "fft", 10000, 100000, False, False, False, 10, 3, True),
        ("float32 device slab with vacf_by_particle: FFT VACF 10000 x 100000 x 3", "fft", 10000, 100000, True, False, False, 3, 1, True),
        # millions of particles, a few dozen frames (k_short, short_kernels.hpp: a lane per column, every lag in its registers)
        ("short trajectories: FFT VACF timeseries 32 x 15624960 x 3 (12 GB)", "fft", 32, 15624960, False, False, False, 5, 2),
        ("short trajectories with vacf_by_particle (the array written in place): FFT VACF 32 x 15624960 x 3", "fft", 32, 15624960, True, False, False, 3, 1),
    ]
    for spec in specs:
        name, mode, T, A, byp, f32, hfft, steps, warm = spec[:9]
        slab32 = len(spec) > 9 and spec[9]
        try:
            ctx.stage_free()
            ctx.trim()
            torch.cuda.empty_cache()
            seed = SEED + {"fft": 3, "direct": 4, "helfand": 5}[mode]
            c = Case(torch, ctx, dev, mode, T, A, 3, 0, A * 3, seed, byp, f32, hfft, slab32)
            el = timed(torch, dist, 1, steps, warm, c.step)
            hist = ctx.timing_history(steps)
            kms = statistics.median(m for _, m in hist)
            tms = statistics.median(t for t, _ in hist)
            composite = byp or hfft  # several kernels share the work: price the whole call
            r = roofline_of(c, tms if composite else kms, hfft, f32)
            if composite:
                r["kernel"] = "whole call"
            r["traffic"] = recorded_traffic(f"{mode}_{T}x{A}x3" + ("_bp" if byp else "") + ("_hfft" if hfft else "")
                                            + ("_f32" if f32 else "") + ("_slab32" if slab32 else ""))
            r["kernels"] = kernel_split(ctx, c, torch)
            # the same bytes / flops over the dominant kernel alone and over the whole call (every launch of the path)
            used = tms if composite else kms
            r["frac_kernel"], r["frac_call"] = r["frac"] * used / kms, r["frac"] * used / tms
            res.append({"workload": name, "ms_per_step": el / steps * 1e3, "steps": steps,
                        "value": T * A / (el / steps), "unit": "lag-points/s",
                        "device_ms": {"whole_call_median": tms, "dominant_kernel_median": kms}, "roofline": r})
            del c
        except Exception as e:
            res.append({"workload": name, "error": str(e)[:300]})
    ctx.set_option("direct_f32", 0)
    ctx.set_option("stage_device_f32", 0)
    ctx.set_option("helfand_fft", 0)
    return res


if __name__ == "__main__":
    main()
