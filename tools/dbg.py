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
    print(T, A, D, "M", _lib.fft_plan_info(T)["M"], "nwg", nwg, "err %.2e" % e, flush=True)
for T in (600, 1100, 1500, 2049, 2561, 3000, 4097, 9000):
    for (A, D) in ((4, 2), (3, 3), (64, 3), (600, 2)):
        run(T, A, D)
