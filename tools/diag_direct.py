"""Sweep of the direct correlators' launch shape on the GPU box (diagnostic):
   python tools/diag_direct.py [T A] -- kernel ms per (chunk L, groups G) through the staged API."""
import sys

import torch

sys.path.insert(0, ".")
from transport_analysis_amd import _lib  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
A = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
mode = sys.argv[3] if len(sys.argv) > 3 else "vacf"
ctx = _lib.Context(0)
st = torch.cuda.current_stream().cuda_stream
ctx.stage_alloc_device(T, A, 3, n_slabs=2 if mode == "helfand" else 1)
ctx.stage_synth(0, 1, 0, A * 3, st)
if mode == "helfand":
    ctx.stage_synth(1, 2, 0, A * 3, st)
m = torch.ones(A, dtype=torch.float64, device="cuda")
out = torch.zeros(T, dtype=torch.float64, device="cuda")
for f32 in ((0, 1) if mode == "helfand" else (0,)):
    ctx.set_option("direct_f32", f32)
    for L in (8, 10):
        for G in (0, 1, 2, 3, 4, 5):
            ctx.set_option("direct_chunk", L)
            ctx.set_option("direct_groups", G)
            for _ in range(3):
                if mode == "helfand":
                    ctx.helfand_msd_staged(m.data_ptr(), 1.0, out.data_ptr(), 0, A, st)
                else:
                    ctx.vacf_direct_staged(out.data_ptr(), 0, A, st)
            torch.cuda.synchronize()
            print(f"{mode} f32={f32} T={T} A={A} L={L} G={G or 'auto'}: kernel {ctx.last_timing()[1]:.2f} ms", flush=True)
