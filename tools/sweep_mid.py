#!/usr/bin/env python3
"""The O(T^2) analyses between the short kernels (64 frames) and the lengths where the matrix-core kernels are efficient, at
equal data volume (n_frames x n_atoms = 5e8: 12 GB of float64 input per slab): the default choice ("direct_mfma" 1), the
vector kernel with column groups of 16 / 32 lanes ("direct_mfma" 0), the same with whole-wave groups ("direct_subwave" 0),
the matrix-core kernels forced ("direct_mfma" 3).     tools/sweep_mid.py -> profiles/r06_direct_mid_sweep.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from transport_analysis_amd import _lib


MID_DEFAULT = 512  # the library's "mid_max" default


def main():
    dev = torch.device("cuda:0")
    ctx = _lib.Context(0)
    frames = [int(a) for a in sys.argv[1:]] or [65, 96, 128, 200, 256, 384, 512, 768, 1000, 1500]
    print("# ms per call (best of 2 after a warm-up): default / vector kernel, column groups of 8-32 lanes, 8 / 10 lags per chunk / vector kernel, whole-wave groups / matrix cores forced / k_mid (\"mid_max\" 512; F32=1: the float32 option, no k_mid)")
    f32 = os.environ.get("F32") == "1"
    for mode in (("helfand",) if f32 else ("direct", "helfand")):
        for bp in (True, False):
            for T in frames:
                A = int(5e8 / T) // 64 * 64
                ctx.stage_free()
                ctx.trim()
                torch.cuda.empty_cache()
                c = bench.Case(torch, ctx, dev, mode, T, A, 3, 0, A * 3, bench.SEED + 4, bp, f32, False, False)
                row = []
                for form, sub, chunk in ((1, 1, 0), (0, 1, 8), (0, 1, 10), (0, 0, 0), (3, 1, 0), (1, 1, -1)):
                    ctx.set_option("mid_max", 512 if (chunk < 0 or (form, sub, chunk) == (1, 1, 0)) and not f32 else 0)
                    ctx.set_option("mid_all", 1 if chunk < 0 else 0)
                    chunk = max(chunk, 0)
                    ctx.set_option("direct_mfma", form)
                    ctx.set_option("direct_subwave", sub)
                    ctx.set_option("direct_chunk", chunk)
                    ts = []
                    for r in range(3):
                        torch.cuda.synchronize()
                        c.step()
                        torch.cuda.synchronize()
                        ts.append(ctx.last_timing()[0])
                    row.append(min(ts[1:]))
                ctx.set_option("direct_mfma", 1)
                ctx.set_option("direct_subwave", 1)
                ctx.set_option("direct_chunk", 0)
                ctx.set_option("mid_max", MID_DEFAULT)
                ctx.set_option("mid_all", 0)
                del c
                print(f"{mode:8s} by_particle={int(bp)} T={T:5d} A={A:8d}: " + " / ".join(f"{x:9.3f}" for x in row), flush=True)


if __name__ == "__main__":
    main()
