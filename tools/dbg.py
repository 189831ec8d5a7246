import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import numpy_oracle as orc
from transport_analysis_amd import _lib
ctx = _lib.Context(0)
def run(T, A, D, nwg=0):
    ctx.set_option("fft_nwg", nwg)
    v = orc.synthetic_velocities(T, A, D, seed=1000 + T)
    (slab,) = ctx.stage_alloc(T, A, D, n_slabs=1)
    slab[...] = v
    ctx.stage_commit(0, T)
    ts, _ = ctx.vacf_fft(by_particle=False)
    _, want = orc.vacf_fft_batched(v)
    e = np.max(np.abs(ts - want)) / np.max(np.abs(want))
    print(T, A, D, "nwg", nwg, "err", e, flush=True)
for (T, A, D) in [(5, 2, 2), (17, 4, 1), (64, 5, 2), (33, 9, 2), (16, 4, 2), (64, 64, 2), (5, 20, 2), (200, 4, 2), (1000,4,2)]:
    for nwg in (0, 2, 4, 16, 32):
        run(T, A, D, nwg)
