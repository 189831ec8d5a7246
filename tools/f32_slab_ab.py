"""FFT VACF on float32 device slabs ("stage_device_f32": the forward kernel reads 8-byte rows and
widens them) against float64 slabs of the same values: bit-equal lag sums, and the time per call.

    python tools/f32_slab_ab.py [n_frames] [n_atoms] [dim] [by_particle 0|1]
"""
import statistics
import sys

import numpy as np
import torch

from transport_analysis_amd._lib import Context


def run(T, A, D, f32, byp, frames_f32):
    ctx = Context(0)
    ctx.set_option("stage_device_f32", 1 if f32 else 0)
    ctx.stage_alloc_device(T, A, D)
    st = torch.cuda.current_stream().cuda_stream
    ctx.stage_commit_dev(0, frames_f32.data_ptr(), A * D, 0, T, dtype=np.float32, stream=st)
    lag = torch.zeros(T, dtype=torch.float64, device="cuda")
    bp = torch.empty((T, A), dtype=torch.float64, device="cuda") if byp else None
    d_bp = bp.data_ptr() if byp else 0
    for _ in range(2):
        ctx.vacf_fft_staged(lag.data_ptr(), d_bp, A, st)
    torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.vacf_fft_staged(lag.data_ptr(), d_bp, A, st)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    out = lag.cpu().numpy(), (bp[:, : min(A, 512)].cpu().numpy() if byp else None)
    ctx.close()
    return statistics.median(ms), out


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    A = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    byp = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
    g = torch.Generator(device="cuda").manual_seed(7)
    frames = torch.randn((T, A * D), dtype=torch.float32, device="cuda", generator=g)
    m64, (l64, b64) = run(T, A, D, False, byp, frames)
    m32, (l32, b32) = run(T, A, D, True, byp, frames)
    same = bool(np.array_equal(l64, l32)) and (not byp or bool(np.array_equal(b64, b32)))
    if not same:  # is either of them reproducible at all?
        _, (l64b, _) = run(T, A, D, False, byp, frames)
        _, (l32b, _) = run(T, A, D, True, byp, frames)
        print("  float64 again equal:", bool(np.array_equal(l64, l64b)), " float32 again equal:", bool(np.array_equal(l32, l32b)),
              " max |diff| / max:", float(np.max(np.abs(l64 - l32)) / np.max(np.abs(l64))), flush=True)
    print(f"{T} x {A} x {D} by_particle={int(byp)}: float64 slab {m64:.3f} ms, float32 slab {m32:.3f} ms "
          f"({m64 / m32:.3f}x), results bit-equal: {same}", flush=True)


if __name__ == "__main__":
    main()
