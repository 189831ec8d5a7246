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


def main(n_cases, seed=1234):
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    st = torch.cuda.current_stream().cuda_stream
    worst = 0.0
    edges = [16, 20, 32, 40, 64, 80, 128, 160, 256, 320, 512, 640, 1024, 1280, 2048, 2560, 4096,
             5120, 8192, 10240]
    for case in range(n_cases):
        kind = rng.choice(["fft", "fft", "fft", "direct", "helfand"])
        if kind == "fft":
            e = int(rng.choice(edges))
            T = int(np.clip(e + rng.integers(-3, 2), 1, 10240)) if rng.random() < 0.6 else int(rng.integers(1, 10241))
            A_all = int(rng.integers(1, 24 if T > 3000 else 60))
        else:
            T = int(rng.integers(1, 700))
            A_all = int(rng.integers(1, 50))
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
        f32 = int(kind != "fft" and rng.random() < 0.3)
        ctx.set_option("direct_f32", f32)
        lag = torch.full((T,), -3.0, dtype=torch.float64, device="cuda")
        bp = torch.full((T, ld_bp), -7.0, dtype=torch.float64, device="cuda")
        d_bp = bp.data_ptr() if by_particle else 0
        vs, xs = v[:, lo:hi], x[:, lo:hi]
        if kind == "fft":
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
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200,
                  int(sys.argv[2]) if len(sys.argv) > 2 else 1234))
