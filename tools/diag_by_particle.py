#!/usr/bin/env python3
"""Diagnostic: kernel time of the FFT by-particle path (10000 x 20000 x 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transport_analysis_amd import _lib
T, A, D = 10000, 20000, 3
vel = torch.randn((T, A, D), dtype=torch.float64, device="cuda")
out = torch.zeros(T, dtype=torch.float64, device="cuda")
bp = torch.zeros((T, A), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for flags in (0,):
    ctx = _lib.Context(0)
    ctx.set_option("fft_debug", flags)
    best = 1e9
    for _ in range(3):
        ctx.vacf_fft_dev(vel.data_ptr(), T, A, D, A * D, out.data_ptr(), bp.data_ptr(), A, st)
        best = min(best, ctx.last_timing()[1])
    print("flags", flags, "ms", round(best, 2), flush=True)
