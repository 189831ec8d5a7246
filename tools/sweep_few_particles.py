import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from transport_analysis_amd import _lib
dev = torch.device("cuda:0"); ctx = _lib.Context(0)
print("# few particles, long series: ms per call, direct_mfma 3 (time-packed) / 2 (column-packed) / 0 (vector)")
for mode, f32 in (("direct", False), ("helfand", False), ("helfand", True)):
    for bp in (False, True):
        for T, A in ((20000, 16), (20000, 64), (20000, 256), (20000, 1024), (5000, 64), (5000, 512), (5000, 4096)):
            row = []
            for form in (3, 0):
                ctx.stage_free(); ctx.trim(); torch.cuda.empty_cache()
                c = bench.Case(torch, ctx, dev, mode, T, A, 3, 0, A * 3, bench.SEED + 4, bp, f32, False, False)
                ctx.set_option("direct_mfma", form)
                ts = []
                for r in range(4):
                    torch.cuda.synchronize(); c.step(); torch.cuda.synchronize(); ts.append(ctx.last_timing()[0])
                row.append(sorted(ts[1:])[1]); del c
            ctx.set_option("direct_mfma", 1); ctx.set_option("direct_f32", 0)
            print(f"{mode:8s} f32={int(f32)} bp={int(bp)} T={T:5d} A={A:5d}: " + " / ".join(f"{x:8.3f}" for x in row), flush=True)
