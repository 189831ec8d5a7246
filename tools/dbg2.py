import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import numpy_oracle as orc
from transport_analysis_amd import _lib
ctx = _lib.Context(0)
for T in (1500, 2049, 3000):
    A, D = 1, 2
    v = np.zeros((T, A, D)); v[0, 0, 0] = 1.0   # delta: |Z|^2 = 1 for all bins -> acf = [1, 0, 0, ...]/(T-k)
    (slab,) = ctx.stage_alloc(T, A, D, n_slabs=1); slab[...] = v; ctx.stage_commit(0, T)
    ts, _ = ctx.vacf_fft(by_particle=False)
    print(T, "delta:", ts[:6], "nonzero count", np.sum(np.abs(ts) > 1e-12), "max idx", np.argmax(np.abs(ts[1:]))+1, np.max(np.abs(ts[1:])))
    v = np.zeros((T, A, D)); v[:, 0, 0] = 1.0     # constant: acf[k] = 1
    (slab,) = ctx.stage_alloc(T, A, D, n_slabs=1); slab[...] = v; ctx.stage_commit(0, T)
    ts, _ = ctx.vacf_fft(by_particle=False)
    print(T, "const:", ts[:4], ts[-3:], "maxdev", np.max(np.abs(ts - 1)))
