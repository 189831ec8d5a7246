#!/usr/bin/env python3
"""Per-unit cost of the FFT forward kernel in by-particle mode against lag-sum mode at equal unit counts (round 6):
dim = 2 (one complex unit per atom), dim = 3 (a complex and a shared real / mixed unit), dim = 1; 10000 frames, 24 GB."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
from transport_analysis_amd import _lib

dev = torch.device("cuda:0")
ctx = _lib.Context(0)
ctx.set_option("timeline", 1)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for D, A in ((2, 150000), (3, 100000), (1, 300000)):
    for bp in (False, True):
        ctx.stage_free()
        ctx.trim()
        torch.cuda.empty_cache()
        c = bench.Case(torch, ctx, dev, "fft", T, A, D, 0, A * D, bench.SEED + 3, bp, False, False, False)
        for r in range(3):
            c.step()
            torch.cuda.synchronize()
            tl = ctx.kernel_timeline()
        print(f"T={T} A={A} D={D} by_particle={bp}: " + "  ".join(f"{n} {m:.3f}" for n, m in tl), flush=True)
        del c
