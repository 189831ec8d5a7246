"""By-particle FFT evaluation (forward kernel + inverse kernel per block of atoms) timed with
the inverse kernel's spectrum prefetch depth 0..3 on one box; prints the per-call time (HIP
events around the call) and the library's own (total, main kernels) timing of the last call.

    python tools/bp_ab.py [n_frames] [n_atoms] [dim] [spec_atoms]
"""
import sys

import numpy as np
import torch

from transport_analysis_amd._lib import Context


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    A = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    spec_atoms = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.stage_alloc_device(T, A, D)
    ctx.stage_synth(0, 1234, 0, A * D)
    if spec_atoms:
        ctx.set_option("bp_spec_atoms", spec_atoms)
    stream = torch.cuda.current_stream(dev).cuda_stream
    res = {}
    for name, pf in (("prefetch 2", 2), ("prefetch 0", 0), ("prefetch 1", 1), ("prefetch 3", 3), ("prefetch 2", 2)):
        ctx.set_option("bp_prefetch", pf)
        lag = torch.zeros(T, dtype=torch.float64, device=dev)
        bp = torch.empty((T, A), dtype=torch.float64, device=dev)
        for _ in range(2):
            ctx.vacf_fft_staged(lag.data_ptr(), bp.data_ptr(), A, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            ctx.vacf_fft_staged(lag.data_ptr(), bp.data_ptr(), A, stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        tm = ctx.last_timing()
        print(f"{name:15s} {ms:8.3f} ms/call  atoms*frames/s {A * T / ms / 1e-3:.3e}  last_timing {tm}", flush=True)
        res.setdefault(name, (lag.cpu().numpy(), bp[:, : min(A, 4096)].cpu().numpy()))
        del lag, bp
    a, b = res["prefetch 2"], res["prefetch 0"]
    print("max |lag diff|", float(np.max(np.abs(a[0] - b[0]))), "max |bp diff|", float(np.max(np.abs(a[1] - b[1]))),
          "bp scale", float(np.max(np.abs(b[1]))))


if __name__ == "__main__":
    main()
