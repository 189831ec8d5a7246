#!/usr/bin/env python3
"""Randomised GPU stress of the FFT / direct / Helfand device entry points against the NumPy
oracle (not collected by pytest: run it by hand on a GPU box, `python tests/stress_gpu.py N`).
Shapes concentrate on the plan boundaries and on the software-pipeline corner cases: one unit
per workgroup, many units per workgroup, odd column counts, unaligned column blocks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import numpy_oracle as orc
from transport_analysis_amd import _lib


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def host_cases(n_cases, seed):
    """Host-facing boundary: pinned slabs in float32 or float64, committed in random chunks."""
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(n_cases):
        ctx = _lib.Context(0)
        T = int(rng.integers(1, 4000)) if rng.random() < 0.8 else int(rng.integers(4000, 18000))
        A = int(rng.integers(1, 30 if T < 4000 else 6))
        D = int(rng.integers(1, 4))
        dtype = np.float32 if rng.random() < 0.5 else np.float64
        kind = rng.choice(["fft", "direct", "helfand"])
        v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=int(rng.integers(1 << 30)))
        v, x = v.astype(dtype), x.astype(dtype)
        slabs = ctx.stage_alloc(T, A, D, n_slabs=2 if kind == "helfand" else 1, dtype=dtype)
        cuts = sorted(set([0, T] + [int(c) for c in rng.integers(0, T + 1, size=int(rng.integers(0, 4)))]))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            slabs[0][lo:hi] = v[lo:hi]
            if kind == "helfand":
                slabs[1][lo:hi] = x[lo:hi]
            ctx.stage_commit(lo, hi)
        v64, x64 = v.astype(np.float64), x.astype(np.float64)
        by_particle = bool(rng.random() < 0.5)
        f32 = bool(kind == "helfand" and T <= 2500 and rng.random() < 0.4)  # float32 option (FP32 matrix cores without bp)
        ctx.set_option("direct_f32", int(f32))
        tol_host = 2e-6 if f32 else 1e-10
        if kind == "helfand":
            scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
            ts, bp = ctx.helfand_msd(m, scale, by_particle=by_particle)
            if T <= 2500:
                want_bp, want_ts = orc.helfand(v64, x64, m, vol, 300.0)
            else:  # O(T^2) oracle: a few lags only
                P = m[None, :, None] * v64 * x64
                ks = [1, 2, T // 2, T - 1]
                sub = np.array([np.mean(np.square(P[:-k] - P[k:]).mean(axis=-1), axis=0) * scale for k in ks])
                e = float(np.max(np.abs(ts[ks] - sub.mean(axis=1)))) / max(float(np.max(np.abs(sub))), 1e-300)
                worst = max(worst, e / 1e-10)
                assert e < 1e-10, ("helfand long", T, A, D, e)
                ctx.close()
                continue
        else:
            ts, bp = (ctx.vacf_fft if kind == "fft" else ctx.vacf_direct)(by_particle=by_particle)
            want_bp, want_ts = orc.vacf_fft_batched(v64)
        sc = max(float(np.max(np.abs(want_bp))), 1e-300)
        e = float(np.max(np.abs(ts - want_ts))) / sc
        if by_particle:
            e = max(e, float(np.max(np.abs(bp - want_bp))) / sc)
        worst = max(worst, e / tol_host)
        if e > tol_host:
            print("FAIL host", case, kind, T, A, D, dtype, by_particle, f32, e, flush=True)
            return 1
        ctx.close()
    print("host stress ok:", n_cases, "cases, worst err/tol %.3g" % worst)
    return 0


def group_cases(n_cases, seed):
    """Several device members behind one call (ta_group; all on GPU 0): random member counts,
    shapes (more members than atoms included), kinds, staging dtypes and by-particle columns."""
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(n_cases):
        n_dev = int(rng.integers(1, 6))
        T = int(rng.integers(1, 3000)) if rng.random() < 0.8 else int(rng.integers(9000, 12000))
        A = int(rng.integers(1, 40 if T < 3000 else 8))
        D = int(rng.integers(1, 4))
        dtype = np.float32 if rng.random() < 0.5 else np.float64
        kind = rng.choice(["fft", "direct", "helfand"]) if T < 3000 else "fft"
        v, x, m, vol = orc.synthetic_helfand(T, A, D, seed=int(rng.integers(1 << 30)))
        v, x = v.astype(dtype), x.astype(dtype)
        g = _lib.Group([0] * n_dev)
        slabs = g.stage_alloc(T, A, D, n_slabs=2 if kind == "helfand" else 1, dtype=dtype)
        cut = int(rng.integers(0, T + 1))
        for lo_t, hi_t in ((0, cut), (cut, T)):
            for view_v, (lo, hi) in zip(slabs[0], g.shards):
                if hi > lo:
                    view_v[lo_t:hi_t] = v[lo_t:hi_t, lo:hi]
            if kind == "helfand":
                for view_x, (lo, hi) in zip(slabs[1], g.shards):
                    if hi > lo:
                        view_x[lo_t:hi_t] = x[lo_t:hi_t, lo:hi]
            g.stage_commit(lo_t, hi_t)
        v64, x64 = v.astype(np.float64), x.astype(np.float64)
        by_particle = bool(rng.random() < 0.6)
        if kind == "helfand":
            scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
            ts, bp = g.helfand_msd(m, scale, by_particle=by_particle)
            want_bp, want_ts = orc.helfand(v64, x64, m, vol, 300.0)
        else:
            ts, bp = (g.vacf_fft if kind == "fft" else g.vacf_direct)(by_particle=by_particle)
            want_bp, want_ts = orc.vacf_fft_batched(v64)
        active = sum(1 for lo, hi in g.shards if hi > lo)
        assert g.reduce_kind == ("none" if active <= 1 else "peer-copy"), (g.reduce_kind, g.shards)
        sc = max(float(np.max(np.abs(want_bp))), 1e-300)
        e = float(np.max(np.abs(ts - want_ts))) / sc
        if by_particle:
            e = max(e, float(np.max(np.abs(bp - want_bp))) / sc)
        worst = max(worst, e / 1e-10)
        if e > 1e-10:
            print("FAIL group", case, kind, n_dev, T, A, D, dtype, by_particle, e, flush=True)
            return 1
        g.close()
    print("group stress ok:", n_cases, "cases, worst err/tol %.3g" % worst)
    return 0


def main(n_cases, seed=1234):
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    st = torch.cuda.current_stream().cuda_stream
    worst = 0.0
    edges = [16, 20, 32, 40, 64, 80, 128, 160, 256, 320, 512, 640, 1024, 1280, 2048, 2560, 4096,
             5120, 8192, 10240]
    for case in range(n_cases):
        kind = rng.choice(["fft", "fft", "fft", "direct", "helfand", "fftlong"])
        if kind == "fftlong":  # beyond one on-chip transform: outer radix (csrc/wfft.hpp)
            T = int(rng.choice([10241, 16384, 16385, 20480, 20481, 32769, 40961])) if rng.random() < 0.4 \
                else int(rng.integers(10241, 60000))
            A_all = int(rng.integers(1, 7))
        elif kind == "fft":
            e = int(rng.choice(edges))
            T = int(np.clip(e + rng.integers(-3, 2), 1, 10240)) if rng.random() < 0.6 else int(rng.integers(1, 10241))
            A_all = int(rng.integers(1, 24 if T > 3000 else 60 if T > 64 else 400))
        elif rng.random() < 0.3:  # short trajectories (short_kernels.hpp): several waves' worth of particles
            T = int(rng.integers(1, 66))
            A_all = int(rng.integers(1, 400))
        else:  # (the matrix-core band kernels: several 256-lag groups now and then)
            T = int(rng.integers(1, 700)) if rng.random() < 0.8 else int(rng.integers(700, 3000))
            A_all = int(rng.integers(1, 50 if T < 700 else 12))
        D = int(rng.integers(1, 4))
        lo = int(rng.integers(0, A_all))
        hi = int(rng.integers(lo + 1, A_all + 1))
        A = hi - lo
        v, x, m, vol = orc.synthetic_helfand(T, A_all, D, seed=int(rng.integers(1 << 30)))
        dv, dx = torch.from_numpy(v).cuda(), torch.from_numpy(x).cuda()
        dm = torch.from_numpy(m[lo:hi].copy()).cuda()
        ld_row, off = A_all * D, lo * D * 8
        by_particle = bool(rng.random() < 0.5)
        ld_bp = A + int(rng.integers(0, 3))
        ctx.set_option("fft_nwg", int(rng.choice([0, 0, 1, 2, 3, 8, 16])))
        ctx.set_option("direct_nwg", int(rng.choice([0, 0, 1, 2, 5])))
        f32 = int(kind in ("direct", "helfand") and rng.random() < 0.3)
        ctx.set_option("direct_f32", f32)
        ctx.set_option("direct_mfma", int(rng.choice([1, 1, 3, 3, 0])))  # by length / matrix cores always / vector kernels
        hfft = int(kind == "helfand" and not f32 and rng.random() < 0.4)
        ctx.set_option("helfand_fft", hfft)  # S1 - 2 S2 through the FFT lag-sum path
        lag = torch.full((T,), -3.0, dtype=torch.float64, device="cuda")
        bp = torch.full((T, ld_bp), -7.0, dtype=torch.float64, device="cuda")
        d_bp = bp.data_ptr() if by_particle else 0
        vs, xs = v[:, lo:hi], x[:, lo:hi]
        if kind in ("fft", "fftlong"):
            ctx.vacf_fft_dev(dv.data_ptr() + off, T, A, D, ld_row, lag.data_ptr(), d_bp, ld_bp, st)
            want_bp, _ = orc.vacf_fft_batched(vs)
        elif kind == "direct":
            ctx.vacf_direct_dev(dv.data_ptr() + off, T, A, D, ld_row, lag.data_ptr(), d_bp, ld_bp, st)
            want_bp, _ = orc.vacf_windowed(vs)
        else:
            scale = 1.0 / (2 * orc.BOLTZMANN_KJ_PER_MOL_K * np.average(vol) * 300.0)
            ctx.helfand_msd_dev(dv.data_ptr() + off, dx.data_ptr() + off, dm.data_ptr(), T, A, D, ld_row,
                                scale, lag.data_ptr(), d_bp, ld_bp, st)
            want_bp, _ = orc.helfand(vs, xs, m[lo:hi], vol, 300.0)
        torch.cuda.synchronize()
        tol = 2e-6 if f32 else 1e-10
        scale_ref = max(np.max(np.abs(want_bp)), 1e-300)
        e1 = float(np.max(np.abs(lag.cpu().numpy() - want_bp.sum(axis=1)))) / (scale_ref * A)
        e2 = 0.0
        if by_particle:
            got = bp.cpu().numpy()
            e2 = float(np.max(np.abs(got[:, :A] - want_bp))) / scale_ref
            assert np.all(got[:, A:] == -7.0), "padding columns touched"
        worst = max(worst, e1 / tol, e2 / tol)
        if e1 > tol or e2 > tol:
            print("FAIL", case, kind, T, A_all, D, lo, hi, by_particle, f32, e1, e2, flush=True)
            return 1
        if case % 25 == 0:
            print("case", case, kind, T, A, D, "ok; worst err/tol so far %.3g" % worst, flush=True)
    print("stress ok:", n_cases, "cases, worst err/tol %.3g" % worst)
    return 0


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
    sys.exit(main(n, sd) or host_cases(max(1, n // 5), sd + 1) or group_cases(max(1, n // 5), sd + 2))
