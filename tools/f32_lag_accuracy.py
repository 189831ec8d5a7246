import numpy as np, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from transport_analysis_amd import ViscosityHelfand as VH
from transport_analysis_amd._mini_mda import ArrayUniverse
t = np.arange(5001, dtype=np.float64)
v = np.repeat(t[:, None, None], 3, axis=2); x = np.repeat((t * t / 2)[:, None, None], 3, axis=2)
u = ArrayUniverse(positions=x, velocities=v, masses=[16.0], dimensions=[2, 2, 2, 90, 90, 90])
want = np.load("/root/repo/tests/golden/kat_helfand_poly_10_1000_10_D2.npy")
vh = VH(u.atoms, dim_type="xy", float32=True).run(start=10, stop=1000, step=10)
rel = np.abs(vh.results.timeseries[1:] - want[1:]) / np.abs(want[1:])
for lo, hi in ((1, 2), (2, 5), (5, 10), (10, 30), (30, 99)):
    print(lo, hi, rel[lo - 1:hi - 1].max())
