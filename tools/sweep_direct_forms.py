#!/usr/bin/env python3
"""The O(T^2) analyses at short and long trajectories, equal work (T^2 x A fixed): time-packed matrix-core kernels
("direct_mfma" 1) against the vector kernels (0) and the column-packed matrix-core forms (2).
    tools/sweep_direct_forms.py  -> profiles/r05_direct_forms_sweep.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from transport_analysis_amd import _lib


def main():
    dev = torch.device("cuda:0")
    ctx = _lib.Context(0)
    print("# ms per call (median of 3 after a warm-up); columns: direct_mfma 1 (time-packed) / 0 (vector) / 2 (column-packed)")
    for mode, f32 in (("direct", False), ("helfand", False), ("helfand", True)):
        for bp in (True, False):
            for T in ([int(x) for x in os.environ['SWEEP_T'].split(',')] if os.environ.get('SWEEP_T') else (64, 200, 500, 1000, 2000, 5000)):
                A = max(64, int(5000 * 5000 * 20000 / (T * T)) // 64 * 64)
                A = min(A, int(20e9 / (T * 3 * 8 * (2 if mode == "helfand" else 1) + (T * 8 * 2 if bp else 0))))
                row = []
                for form in (1, 0):
                    ctx.stage_free()
                    ctx.trim()
                    torch.cuda.empty_cache()
                    c = bench.Case(torch, ctx, dev, mode, T, A, 3, 0, A * 3, bench.SEED + 4, bp, f32, False, False)
                    ctx.set_option("direct_mfma", form)
                    ts = []
                    for r in range(4):
                        torch.cuda.synchronize()
                        c.step()
                        torch.cuda.synchronize()
                        ts.append(ctx.last_timing()[0])
                    row.append(sorted(ts[1:])[1])
                    del c
                ctx.set_option("direct_mfma", 1)
                ctx.set_option("direct_f32", 0)
                print(f"{mode:8s} f32={int(f32)} by_particle={int(bp)} T={T:5d} A={A:8d}: " + " / ".join(f"{x:9.3f}" for x in row), flush=True)


if __name__ == "__main__":
    main()
