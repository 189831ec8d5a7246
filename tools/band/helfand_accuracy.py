"""Matrix-core Einstein-Helfand against the difference-first vector kernel ("direct_mfma" 0), lag by lag (and particle by
particle): the time-packed kernels ("direct_mfma" 3: k_band_bp_helf, k_band32_tp) and the column-packed ones (2).
Random-walk positions with an offset and a drift; a pure cubic trend.  -> profiles/r05_band_helfand_accuracy.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from oracle import numpy_oracle as orc  # noqa: E402
from transport_analysis_amd import _lib  # noqa: E402

ctx = _lib.Context(0)


def run(v, x, m, scale, form, bp, f32=False):
    T, A, D = v.shape
    sv, sx = ctx.stage_alloc(T, A, D, n_slabs=2)
    sv[...] = v
    sx[...] = x
    ctx.stage_commit(0, T)
    ctx.set_option("direct_f32", int(f32))
    ctx.set_option("direct_mfma", form)
    a = ctx.helfand_msd(m, scale, by_particle=bp)[1 if bp else 0].copy()
    ctx.set_option("direct_f32", 0)
    ctx.set_option("direct_mfma", 0)
    b = ctx.helfand_msd(m, scale, by_particle=bp)[1 if bp else 0].copy()
    ctx.set_option("direct_mfma", 1)
    return a, b


def report(tag, a, b):
    rel = np.abs(a[1:] - b[1:]) / np.abs(b[1:])
    k = np.unravel_index(rel.argmax(), rel.shape)
    print("%-92s lag by lag %.2e (lag %d), lag 1 %.2e, of the scale %.2e" % (tag, rel.max(), k[0] + 1, rel[0].max(), np.abs(a - b).max() / np.abs(b).max()))


for form, name in ((3, "time-packed"), (2, "column-packed")):
    for T, A in ((3000, 40), (5000, 7), (20000, 3)):
        v, x, m, vol = orc.synthetic_helfand(T, A, 3, seed=5)
        x = x + 50.0 + 0.01 * np.arange(T)[:, None, None]
        for bp in (False, True):
            if form == 2 and bp:
                continue  # (no column-packed float64 by-particle form)
            report("float64 %s, random walk + offset + drift, T=%d A=%d, %s:" % (name, T, A, "by particle" if bp else "lag sums"), *run(v, x, m, 1.0, form, bp))
        report("float32 %s, the same, lag sums (bar: 2e-6 of the scale):" % name, *run(v, x, m, 1.0, form, False, True))
    T = 3000
    t = np.arange(T, dtype=np.float64)
    v = np.repeat(t[:, None, None], 3, axis=2) * np.ones((1, 2, 1))
    v[:, 1] *= 0.5
    x = np.repeat((t * t / 2)[:, None, None], 3, axis=2) * np.ones((1, 2, 1))
    for bp in (False, True):
        if form == 2 and bp:
            continue
        report("float64 %s, pure trend P = m t^3 / 2, T=3000, %s:" % (name, "by particle" if bp else "lag sums"), *run(v, x, np.array([1.0, 2.0]), 1.0, form, bp))
    report("float32 %s, pure trend, lag sums:" % name, *run(v, x, np.array([1.0, 2.0]), 1.0, form, False, True))
