"""INTEGRATION.md section B -- the binding a reference maintainer would add -- executed AS WRITTEN.

The python block is cut out of the document and exec'd in a fresh interpreter that has not loaded
this package's ctypes layer (`transport_analysis_amd._lib` never opens the library there): raw
`ctypes.CDLL("libta_hip.so")`, found through LD_LIBRARY_PATH.  The driver supplies only what the
reference itself supplies around the three hooks: `AnalysisBase`, `NoDataError` and the
constructor's attributes (/root/reference/transport_analysis/velocityautocorr.py:112-140).  The
result is compared with the oracle at the parity bar of tests/test_gpu_parity.py.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, scale_rel_err

TOL = 1e-10

DRIVER = r'''
import sys, numpy as np
sys.path.insert(0, {repo!r})
from transport_analysis_amd._base import AnalysisBase, NoDataError     # the MDAnalysis protocol (stand-in)
from transport_analysis_amd._mini_mda import ArrayUniverse
ns = dict(AnalysisBase=AnalysisBase, NoDataError=NoDataError)
exec(compile(open({snippet!r}).read(), "INTEGRATION.md#B", "exec"), ns)


class Patched(ns["VelocityAutocorr"]):
    def __init__(self, atomgroup, fft=True):     # what velocityautocorr.py:112-140 sets for dim_type="xyz"
        super().__init__(atomgroup.universe.trajectory)
        self.atomgroup, self.fft = atomgroup, fft
        self._dim, self.dim_fac, self.n_particles = [0, 1, 2], 3, len(atomgroup)


v = np.load({vel!r})
out = {{}}
for fft in (True, False):
    a = Patched(ArrayUniverse(velocities=v).atoms, fft=fft).run()
    out[f"ts{{int(fft)}}"], out[f"bp{{int(fft)}}"] = a.results.timeseries, np.array(a.results.vacf_by_particle)
    del a
import gc; gc.collect()
# the package's own ctypes layer never loaded the library: everything above went through the snippet's CDLL
from transport_analysis_amd import _lib as _pkg_binding
assert _pkg_binding._lib is None, "the snippet must not lean on this package's binding"
np.savez({out!r}, **out)
'''


def snippet_text():
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    sec = doc[doc.index("## B. The binding a reference maintainer would add"):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "section B has no python block"
    return m.group(1)


def test_snippet_is_raw_ctypes():
    """(CPU) the documented binding uses nothing but ctypes + numpy and frees what it allocates."""
    src = snippet_text()
    compile(src, "INTEGRATION.md#B", "exec")
    assert 'ctypes.CDLL("libta_hip.so")' in src and "transport_analysis_amd" not in src
    assert "ta_host_free" in src and "ta_ctx_destroy" in src


@pytest.mark.gpu
def test_integration_snippet_runs_as_written(tmp_path):
    from oracle import numpy_oracle as orc

    T, A = 700, 37
    rng = np.random.default_rng(2025)
    # float32-representable values: the trajectory hands out float32 copies, as MDAnalysis does
    v = rng.standard_normal((T, A, 3)).astype(np.float32).astype(np.float64)
    np.save(tmp_path / "v.npy", v)
    (tmp_path / "snippet.py").write_text(snippet_text())
    (tmp_path / "driver.py").write_text(DRIVER.format(repo=REPO, snippet=str(tmp_path / "snippet.py"),
                                                      vel=str(tmp_path / "v.npy"), out=str(tmp_path / "out.npz")))
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(REPO, "transport_analysis_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([sys.executable, str(tmp_path / "driver.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(tmp_path / "out.npz")
    bp_ref, ts_ref = orc.vacf_windowed(v)
    for fft in (1, 0):
        assert scale_rel_err(z[f"ts{fft}"], ts_ref) < TOL
        assert scale_rel_err(z[f"bp{fft}"], bp_ref) < TOL
