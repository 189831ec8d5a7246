"""By-particle FFT evaluation against the atoms per block of power spectra ("bp_spec_atoms"):
does a block whose spectra and atom-major lags fit the 256 MB Infinity Cache run faster than the
2.5 GiB default?  Per setting: ms per call (median of n) and the library's
per-kernel timeline of one extra call.

    python tools/bp_sweep.py [n_frames] [n_atoms] [dim] [settings, comma separated; 0 = default]
"""
import statistics
import sys

import torch

from transport_analysis_amd._lib import Context


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    A = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    settings = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,8192,4096,2048,1024,512,256").split(",")]
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.stage_alloc_device(T, A, D)
    ctx.stage_synth(0, 1234, 0, A * D)
    stream = torch.cuda.current_stream(dev).cuda_stream
    lag = torch.zeros(T, dtype=torch.float64, device=dev)
    bp = torch.empty((T, A), dtype=torch.float64, device=dev)
    for sa in settings:
        ctx.set_option("bp_spec_atoms", sa)
        ctx.set_option("timeline", 0)
        for _ in range(2):
            ctx.vacf_fft_staged(lag.data_ptr(), bp.data_ptr(), A, stream)
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.vacf_fft_staged(lag.data_ptr(), bp.data_ptr(), A, stream)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        ctx.set_option("timeline", 1)
        ctx.vacf_fft_staged(lag.data_ptr(), bp.data_ptr(), A, stream)
        torch.cuda.synchronize()
        tl = ", ".join(f"{n} {v:.2f}" for n, v in ctx.kernel_timeline())
        print(f"bp_spec_atoms {sa:6d}: median {statistics.median(ms):7.3f} ms  min {min(ms):7.3f}   [{tl}]", flush=True)


if __name__ == "__main__":
    main()
