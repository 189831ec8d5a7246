#!/usr/bin/env python3
"""vacf_by_particle (and the lag sums) through the FFT path and through the O(T^2) matrix-core path at equal data volume
(n_frames x n_atoms = 5e8: 12 GB of float64 input): where does the O(T^2) kernel beat the FFT?
    tools/sweep_fft_vs_direct.py -> profiles/r05_fft_vs_direct.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from transport_analysis_amd import _lib


def main():
    dev = torch.device("cuda:0")
    ctx = _lib.Context(0)
    print("# ms per call (median of 3 after a warm-up): FFT path / O(T^2) path (\"direct_mfma\" 1)")
    for bp in (True, False):
        for T in (32, 64, 128, 256, 384, 512, 768, 1024, 2048):
            A = int(5e8 / T) // 64 * 64
            row = []
            for mode in ("fft", "direct"):
                ctx.stage_free()
                ctx.trim()
                torch.cuda.empty_cache()
                c = bench.Case(torch, ctx, dev, mode, T, A, 3, 0, A * 3, bench.SEED + 4, bp, False, False, False)
                ts = []
                for r in range(4):
                    torch.cuda.synchronize()
                    c.step()
                    torch.cuda.synchronize()
                    ts.append(ctx.last_timing()[0])
                row.append(sorted(ts[1:])[1])
                del c
            print(f"by_particle={int(bp)} T={T:5d} A={A:9d}: fft {row[0]:9.3f}  direct {row[1]:9.3f}", flush=True)


if __name__ == "__main__":
    main()
