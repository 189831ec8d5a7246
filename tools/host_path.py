"""Host-facing path timing on the GPU box: pinned slab -> commit -> compute -> results on the host.
   python tools/host_path.py [T A by_particle(0/1) bp_block]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from transport_analysis_amd import _lib  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
A = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
byp = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
ctx = _lib.Context(0)
if len(sys.argv) > 4:
    ctx.set_option("bp_block", int(sys.argv[4]))
(slab,) = ctx.stage_alloc(T, A, 3, n_slabs=1, dtype=np.float32)
rng = np.random.default_rng(5)
blk = rng.standard_normal((200, A, 3), dtype=np.float32)
for t in range(0, T, 200):
    slab[t:t + 200] = blk[: min(200, T - t)]
for rep in range(3):
    t0 = time.perf_counter()
    ctx.stage_commit(0, T)
    t1 = time.perf_counter()
    ts, bp = ctx.vacf_fft(by_particle=byp)
    t2 = time.perf_counter()
    print(f"T={T} A={A} by_particle={byp}: commit call {1e3 * (t1 - t0):.1f} ms, compute+copy {1e3 * (t2 - t1):.1f} ms, "
          f"total {1e3 * (t2 - t0):.1f} ms -> {T * A / (t2 - t0):.3g} lag-points/s", flush=True)
if byp:
    print("check", float(np.abs(bp.mean(axis=1) - ts).max() / np.abs(ts).max()))
