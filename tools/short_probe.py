#!/usr/bin/env python3
"""Short trajectories (8 ... 64 frames) at equal data volume (n_frames x n_atoms = 5e8: 12 GB of float64 input, 4 GB of
by-particle output): k_short (short_kernels.hpp, "short_max" 64) against the kernels it replaces ("short_max" 0) for the
three quantities, with the by-particle array and lag sums alone; the FFT path's lag sums with "short_lags_max".
    tools/short_probe.py -> profiles/r06_short.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from transport_analysis_amd import _lib


def main():
    dev = torch.device("cuda:0")
    ctx = _lib.Context(0)
    frames = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 48, 64]
    print("# ms per call (median of 3 after a warm-up); GB/s = (input + by-particle output) / time; old = \"short_max\" 0")
    for T in frames:
        A = int(5e8 / T) // 64 * 64
        for mode in ("fft", "direct", "helfand"):
            for bp in (True, False):
                ctx.stage_free()
                ctx.trim()
                torch.cuda.empty_cache()
                c = bench.Case(torch, ctx, dev, mode, T, A, 3, 0, A * 3, bench.SEED + 4, bp, False, False, False)
                row = []
                for short in (64, 0):
                    ctx.set_option("short_max", short)
                    ctx.set_option("short_lags_max", short)
                    ts = []
                    for r in range(4):
                        torch.cuda.synchronize()
                        c.step()
                        torch.cuda.synchronize()
                        ts.append(ctx.last_timing()[0])
                    row.append(sorted(ts[1:])[1])
                ctx.set_option("short_max", 64)
                ctx.set_option("short_lags_max", 48)
                gb = (T * A * 3 * 8 * (2 if mode == "helfand" else 1) + (T * A * 8 if bp else 0)) / 1e9
                print(f"T={T:3d} A={A:9d} {mode:8s} by_particle={int(bp)}: k_short {row[0]:8.3f} ms ({gb / row[0]:6.2f} TB/s)"
                      f"   old {row[1]:9.3f} ms", flush=True)
                del c


if __name__ == "__main__":
    main()
