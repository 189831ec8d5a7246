#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-facing boundary: fill pinned slabs, ta_stage_commit
(H2D), ta_vacf_fft (kernels + D2H of the result).  Not the bench metric: bench.py times
device-resident input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from transport_analysis_amd import _lib

def main(T=10000, A=20000, D=3, dtype=np.float64):
    ctx = _lib.Context(0)
    t0 = time.perf_counter()
    (slab,) = ctx.stage_alloc(T, A, D, n_slabs=1, dtype=dtype)
    t1 = time.perf_counter()
    rng = np.random.default_rng(1)
    row = rng.standard_normal((A, D)).astype(dtype)
    for i in range(T):            # stands in for the per-frame _single_frame copies
        slab[i] = row
    t2 = time.perf_counter()
    ctx.stage_commit(0, T)
    ts, _ = ctx.vacf_fft(by_particle=False)   # blocks: waits for the copies, kernels, D2H
    t3 = time.perf_counter()
    ts2, _ = ctx.vacf_fft(by_particle=False)  # data already resident
    t4 = time.perf_counter()
    gb = slab.nbytes / 1e9
    print(f"{np.dtype(dtype).name} slab {T}x{A}x{D} = {gb:.2f} GB: alloc {t1-t0:.3f} s, host fill {t2-t1:.3f} s, "
          f"commit+compute {t3-t2:.3f} s ({gb/(t3-t2):.1f} GB/s incl. PCIe, {T*A/(t3-t2):.3e} lag-points/s), "
          f"compute only {t4-t3:.4f} s ({T*A/(t4-t3):.3e} lag-points/s)", flush=True)

if __name__ == "__main__":
    main(10000, 20000, 3, np.float64)
    main(10000, 20000, 3, np.float32)
