#!/usr/bin/env python3
"""k_mid launch shapes: ms per call at 12 GB for lanes-per-block-pair 64 / 32 / 16 / 8 ("mid_ncl" 6..3)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from transport_analysis_amd import _lib

dev = torch.device("cuda:0")
ctx = _lib.Context(0)
ctx.set_option("mid_max", 512)
for mode in ("direct", "helfand"):
    for bp in (True, False):
        for T in [int(a) for a in sys.argv[1:]] or [65, 96, 128, 200, 256, 384, 512]:
            A = int(5e8 / T) // 64 * 64
            ctx.stage_free(); ctx.trim(); torch.cuda.empty_cache()
            c = bench.Case(torch, ctx, dev, mode, T, A, 3, 0, A * 3, bench.SEED + 4, bp, False, False, False)
            row = []
            for ncl in (6, 5, 4, 3):
                ctx.set_option("mid_ncl", ncl)
                try:
                    ts = []
                    for r in range(3):
                        torch.cuda.synchronize(); c.step(); torch.cuda.synchronize()
                        ts.append(ctx.last_timing()[0])
                    row.append(min(ts[1:]))
                except Exception as e:
                    row.append(float("nan"))
            ctx.set_option("mid_ncl", 0)
            del c
            print(f"{mode:8s} by_particle={int(bp)} T={T:4d}: " + " / ".join(f"{x:8.3f}" for x in row), flush=True)
