#!/usr/bin/env python3
"""Diagnostic: direct-correlator kernel time vs persistent workgroup count / float32 switch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transport_analysis_amd import _lib

def run(mode, T, A, D, nwg, f32, chunk=0):
    ctx = _lib.Context(0)
    ctx.set_option("direct_nwg", nwg)
    ctx.set_option("direct_f32", f32)
    ctx.set_option("direct_chunk", chunk)
    vel = torch.randn((T, A, D), dtype=torch.float64, device="cuda")
    pos = 30 + 0.002 * torch.cumsum(vel, 0) if mode == "helfand" else None
    m = torch.ones(A, dtype=torch.float64, device="cuda")
    out = torch.zeros(T, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    best = 1e9
    for _ in range(3):
        if mode == "helfand":
            ctx.helfand_msd_dev(vel.data_ptr(), pos.data_ptr(), m.data_ptr(), T, A, D, A * D, 1.0, out.data_ptr(), 0, A, st)
        else:
            ctx.vacf_direct_dev(vel.data_ptr(), T, A, D, A * D, out.data_ptr(), 0, A, st)
        best = min(best, ctx.last_timing()[1])
    return best

if __name__ == "__main__":
    for mode, T, A in (("vacf", 5000, 12800), ("helfand", 5000, 5120), ("helfand", 20000, 2048), ("vacf", 1000, 30000), ("vacf", 10000, 4096)):
        for f32 in (0, 1):
            for chunk in (8, 10, 0):
                try:
                    ms = run(mode, T, A, 3, 0, f32, chunk)
                except Exception as e:
                    ms = float("nan")
                print(f"{mode} T={T} A={A} f32={f32} chunk={chunk}: {ms:.2f} ms", flush=True)
