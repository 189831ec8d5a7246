import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from oracle import numpy_oracle as orc
from transport_analysis_amd import _lib
ctx = _lib.Context(0)
def run(v, x, m, scale):
    T, A, D = v.shape
    sv, sx = ctx.stage_alloc(T, A, D, n_slabs=2)
    sv[...] = v; sx[...] = x
    ctx.stage_commit(0, T)
    ctx.set_option("direct_mfma", 1)
    a, _ = ctx.helfand_msd(m, scale, by_particle=False)
    ctx.set_option("direct_mfma", 0)
    b, _ = ctx.helfand_msd(m, scale, by_particle=False)
    ctx.set_option("direct_mfma", 1)
    return a.copy(), b.copy()
for T, A in ((3000, 40), (5000, 7)):
    v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=5)
    x = x + 50.0 + 0.01 * np.arange(T)[:, None, None]
    a, b = run(v, x, m, 1.0)
    rel = np.abs(a[1:] - b[1:]) / np.abs(b[1:])
    print("random walk positions + offset + drift, T=%d A=%d: max lag-by-lag relative deviation %.2e at lag %d; scale-relative %.2e" % (T, A, rel.max(), rel.argmax() + 1, np.abs(a - b).max() / np.abs(b).max()))
T = 3000
t = np.arange(T, dtype=np.float64)
v = np.repeat(t[:, None, None], 3, axis=2) * np.ones((1, 2, 1)); v[:, 1] *= 0.5
x = np.repeat((t * t / 2)[:, None, None], 3, axis=2) * np.ones((1, 2, 1))
a, b = run(v, x, np.array([1.0, 2.0]), 1.0)
rel = np.abs(a[1:] - b[1:]) / np.abs(b[1:])
print("pure trend P = m t^3 / 2, T=3000: max lag-by-lag relative deviation %.2e at lag %d (lag 1: %.2e)" % (rel.max(), rel.argmax() + 1, rel[0]))
