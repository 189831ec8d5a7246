#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Run in the BUILD container only (it reads /root/reference, which does not exist
on the GPU box); the outputs are committed, the reference source is not.

What is executed
----------------
* The reference's own known-answer generators, taken from its test modules by
  AST (no MDAnalysis needed for ``characteristic_poly``;
  ``characteristic_poly_helfand`` is fed a tiny in-memory trajectory object):
  ``transport_analysis/tests/test_velocityautocorr.py:79-93`` and
  ``transport_analysis/tests/test_viscosity.py:89-132``.
* The reference's own ``VelocityAutocorr._conclude_simple`` /
  ``_conclude_fft`` and ``ViscosityHelfand._conclude`` method bodies
  (``velocityautocorr.py:208-238``, ``viscosity.py:201-245``), imported from
  /root/reference and called on a bare namespace object that carries the
  arrays the hooks would have staged.  MDAnalysis and tidynamics are not
  installed in the container, so the import is satisfied with an empty
  ``MDAnalysis`` module tree (class placeholders only, no arithmetic) and with
  ``tidynamics.acf`` bound to the restatement in ``oracle/numpy_oracle.py``:
  the windowed-VACF and Helfand vectors are therefore produced by reference
  code alone; the FFT vectors by reference code + the restated tidynamics, and
  they are labelled as such in ``manifest.json``.
* Constants printed in the reference's tests/notebooks (Green-Kubo numbers,
  the N=10 notebook vectors, the NCBOX water vectors whose input files are not
  available here) are copied as data into ``reference_constants.json``.
"""

import ast
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import numpy_oracle as orc  # noqa: E402


# ---------------------------------------------------------------- helpers
def _function_from(path, name, extra_globals):
    """Compile one top-level function of a reference file, in isolation."""
    with open(path, "r", encoding="utf8") as fh:
        tree = ast.parse(fh.read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = dict(extra_globals)
            exec(compile(mod, path, "exec"), ns)
            return ns[name]
    raise KeyError(name)


def _import_reference_classes():
    """Import the reference package with placeholder third-party modules."""

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class AnalysisBase:  # placeholder: the hooks under test never call it
        def __init__(self, *a, **k):
            pass

    class UpdatingAtomGroup:
        pass

    class NoDataError(ValueError):
        pass

    mda = module("MDAnalysis")
    mda.analysis = module("MDAnalysis.analysis")
    mda.analysis.base = module("MDAnalysis.analysis.base", AnalysisBase=AnalysisBase)
    mda.core = module("MDAnalysis.core")
    mda.core.groups = module(
        "MDAnalysis.core.groups", UpdatingAtomGroup=UpdatingAtomGroup, AtomGroup=object
    )
    mda.exceptions = module("MDAnalysis.exceptions", NoDataError=NoDataError)
    mda.units = module(
        "MDAnalysis.units",
        constants={"Boltzmann_constant": orc.BOLTZMANN_KJ_PER_MOL_K},
    )
    module("tidynamics", acf=orc.tidynamics_acf)
    sys.path.insert(0, REF)
    from transport_analysis.velocityautocorr import VelocityAutocorr
    from transport_analysis.viscosity import ViscosityHelfand

    return VelocityAutocorr, ViscosityHelfand


class _Bag(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def run_ref_vacf(VACF, velocities, fft):
    n_frames, n_particles, _ = velocities.shape
    me = types.SimpleNamespace(
        n_frames=n_frames,
        n_particles=n_particles,
        _velocities=np.array(velocities, dtype=np.float64),
        results=_Bag(vacf_by_particle=np.zeros((n_frames, n_particles))),
        fft=fft,
    )
    (VACF._conclude_fft if fft else VACF._conclude_simple)(me)
    return me.results.vacf_by_particle, me.results.timeseries


def run_ref_helfand(VH, v, x, masses, volumes, temp_avg, fit_window=None):
    n_frames, n_particles, _ = v.shape
    me = types.SimpleNamespace(
        n_frames=n_frames,
        n_particles=n_particles,
        _velocities=np.array(v, dtype=np.float64),
        _positions=np.array(x, dtype=np.float64),
        _volumes=np.array(volumes, dtype=np.float64),
        _masses_rs=np.asarray(masses, dtype=np.float64).reshape((1, n_particles, 1)),
        boltzmann=orc.BOLTZMANN_KJ_PER_MOL_K,
        temp_avg=temp_avg,
        linear_fit_window=fit_window,
        results=_Bag(visc_by_particle=np.zeros((n_frames, n_particles))),
    )
    VH._conclude(me)
    return me.results


def step_trajectory(nstep, start=0, stop=None, step=1):
    """The reference's 'step' trajectory: 1 atom, v=(t,t,t), x=t^2/2
    (test_velocityautocorr.py:46-57, test_viscosity.py:56-86), sliced the way
    AnalysisBase.run(start, stop, step) slices frames."""
    t = np.arange(nstep, dtype=np.float64)[start:stop:step]
    v = np.repeat(t[:, None, None], 3, axis=2)
    x = np.repeat((t * t / 2)[:, None, None], 3, axis=2)
    return v, x


# ------------------------------------------------------------------- main
OUT = HERE  # --check writes to a temporary directory instead


def _docstring_vector(path, after, n):
    """The n numbers of the first array(...) literal printed after the marker text `after`."""
    import re

    text = open(path, "r", encoding="utf8").read()
    i = text.index(after)
    j = text.index("array(", i)
    body = text[j + 6:text.index(")", j)]
    vals = [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body)]
    assert len(vals) == n, (path, after, vals)
    return vals


def main():
    manifest = {}
    ref_tests = os.path.join(REF, "transport_analysis", "tests")

    poly = _function_from(
        os.path.join(ref_tests, "test_velocityautocorr.py"), "characteristic_poly", {"np": np}
    )
    poly_h = _function_from(
        os.path.join(ref_tests, "test_viscosity.py"),
        "characteristic_poly_helfand",
        {"np": np, "constants": {"Boltzmann_constant": orc.BOLTZMANN_KJ_PER_MOL_K}},
    )
    VACF, VH = _import_reference_classes()

    def save(name, arr, source):
        np.save(os.path.join(OUT, name), np.asarray(arr))
        manifest[name] = source

    # --- VACF step-trajectory polynomial (KAT), full and start/stop/step
    for d in (1, 2, 3):
        save(f"kat_vacf_poly_N5001_D{d}.npy", poly(5001, d),
             "reference characteristic_poly(5001, d) [test_velocityautocorr.py:79-93]")
        save(f"kat_vacf_poly_10_1000_10_D{d}.npy", poly(1000, d, first=10, step=10),
             "reference characteristic_poly(1000, d, first=10, step=10)")
    save("kat_vacf_poly_N10_D3.npy", poly(10, 3), "reference characteristic_poly(10, 3)")

    # --- reference methods on the step trajectory (N=10 and sliced N=99)
    v10, x10 = step_trajectory(10)
    bp, ts = run_ref_vacf(VACF, v10, fft=False)
    save("ref_vacf_windowed_step_N10.npy", ts, "reference _conclude_simple on step trajectory N=10")
    bp, ts = run_ref_vacf(VACF, v10, fft=True)
    save("ref_vacf_fft_step_N10.npy", ts,
         "reference _conclude_fft + restated tidynamics on step trajectory N=10")
    v99, x99 = step_trajectory(5001, 10, 1000, 10)
    for d, cols in ((1, [0]), (2, [0, 2]), (3, [0, 1, 2])):
        bp, ts = run_ref_vacf(VACF, v99[:, :, cols], fft=False)
        save(f"ref_vacf_windowed_step_10_1000_10_D{d}.npy", ts,
             "reference _conclude_simple, frames 10:1000:10 of the step trajectory")

    # --- reference methods on seeded random slabs (by-particle + timeseries)
    for (T, A, D, seed) in ((7, 1, 1, 11), (64, 5, 2, 12), (200, 33, 3, 13)):
        v = orc.synthetic_velocities(T, A, D, seed)
        tag = f"T{T}_A{A}_D{D}"
        np.save(os.path.join(OUT, f"rand_vel_{tag}.npy"), v)
        manifest[f"rand_vel_{tag}.npy"] = f"numpy Philox({seed}) standard_normal (input)"
        bp, ts = run_ref_vacf(VACF, v, fft=False)
        save(f"ref_vacf_windowed_bp_{tag}.npy", bp, "reference _conclude_simple vacf_by_particle")
        save(f"ref_vacf_windowed_ts_{tag}.npy", ts, "reference _conclude_simple timeseries")
        bp, ts = run_ref_vacf(VACF, v, fft=True)
        save(f"ref_vacf_fft_bp_{tag}.npy", bp,
             "reference _conclude_fft + restated tidynamics, vacf_by_particle")
        save(f"ref_vacf_fft_ts_{tag}.npy", ts,
             "reference _conclude_fft + restated tidynamics, timeseries")

    # --- Helfand KAT generator of the reference's tests, on the step trajectory
    class _Ts:
        def __init__(self, v, x):
            self.velocities, self.positions = v, x

    class _Traj:
        def __init__(self, v, x):
            self._v, self._x = v, x

        def __getitem__(self, sl):
            return [_Ts(a, b) for a, b in zip(self._v[sl], self._x[sl])]

    class _U:
        pass

    vfull, xfull = step_trajectory(5001)
    u = _U()
    u.trajectory = _Traj(vfull, xfull)
    for d in (1, 2, 3):
        save(f"kat_helfand_poly_N5001_D{d}.npy", poly_h(u, 5001, d),
             "reference characteristic_poly_helfand(u, 5001, d) [test_viscosity.py:89-132]")
        save(f"kat_helfand_poly_10_1000_10_D{d}.npy", poly_h(u, 1000, d, start=10, step=10),
             "reference characteristic_poly_helfand(u, 1000, d, start=10, step=10)")

    # --- reference ViscosityHelfand._conclude on N=10 step and on random data
    res = run_ref_helfand(VH, v10, x10, [16.0], np.full(10, 8.0), 300.0)
    save("ref_helfand_step_N10.npy", res.timeseries,
         "reference ViscosityHelfand._conclude on step trajectory N=10 (m=16, V=8, T=300)")
    for (T, A, D, seed) in ((9, 1, 1, 21), (50, 6, 2, 22), (120, 17, 3, 23)):
        v, x, m, vol = orc.synthetic_helfand(T, A, D, seed)
        vol = vol * (1.0 + 0.01 * np.sin(np.arange(T)))  # non-constant volume
        tag = f"T{T}_A{A}_D{D}"
        np.savez(os.path.join(OUT, f"rand_helfand_in_{tag}.npz"), v=v, x=x, m=m, vol=vol)
        manifest[f"rand_helfand_in_{tag}.npz"] = f"synthetic_helfand(seed={seed}) (input)"
        res = run_ref_helfand(VH, v, x, m, vol, 313.0, fit_window=(2, T - 2))
        save(f"ref_helfand_bp_{tag}.npy", res.visc_by_particle,
             "reference ViscosityHelfand._conclude visc_by_particle (temp_avg=313)")
        save(f"ref_helfand_ts_{tag}.npy", res.timeseries, "reference ... timeseries")
        save(f"ref_helfand_visc_{tag}.npy", np.array([res.viscosity]),
             "reference ... results.viscosity, linear_fit_window=(2, T-2)")

    # --- numbers printed in the reference's tests / notebooks (data only)
    constants = {
        "gk_trapezoid_step_N5001": 24307638750.0,          # test_velocityautocorr.py:378
        "gk_simpson_step_N5001": 24307638888.888885,       # test_velocityautocorr.py:378
        "notebook_fft_step_N10": [85.5, 80.0, 73.5, 66.0, 57.5, 48.0, 37.5, 26.0, 13.5,
                                  -1.0658141e-14],         # vacf_testing_examples.ipynb cell 5
        "notebook_poly_step_N10": [85.5, 80.0, 73.5, 66.0, 57.5, 48.0, 37.5, 26.0, 13.5, 0.0],
        "ncbox_vacf_fft_O_resid_1_10": [4.22902895e+01, -2.69143315e+00, 6.98534787e-01,
                                        2.97003597e+00, 6.34654795e-01, 1.23400236e+00,
                                        -2.78565552e+00, 7.25828978e-01, -1.14432591e-02,
                                        -6.46827198e+00],  # vacf_testing_examples.ipynb cell 3
        "helfand_notebook_step_N10_sum_over_dims": [0.0, 56426.98120794, 192868.75919739],
        # helfand_dev_toy_system.ipynb (stale: summed over dims; current code = these / 3)
        "boltzmann_kJ_per_mol_K": orc.BOLTZMANN_KJ_PER_MOL_K,
        # BASELINE configs[0], "resname WAT and resid 1-5": the vector the reference's module
        # docstring prints (velocityautocorr.py:39-43), read from the file
        "ncbox_vacf_fft_WAT_resid_1_5": _docstring_vector(
            os.path.join(REF, "transport_analysis", "velocityautocorr.py"), "wat_vacf.results.timeseries", 10),
    }
    with open(os.path.join(OUT, "reference_constants.json"), "w") as fh:
        json.dump(constants, fh, indent=1)
    with open(os.path.join(OUT, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("wrote", len(manifest), "arrays")


def check():
    """Regenerate everything into a temporary directory and compare with the committed files:
    arrays bit for bit, JSON by value.  Returns the list of differing file names."""
    import filecmp
    import tempfile

    global OUT
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        OUT = tmp
        try:
            main()
        finally:
            OUT = HERE
        for name in sorted(os.listdir(tmp)):
            a, b = os.path.join(tmp, name), os.path.join(HERE, name)
            if not os.path.exists(b):
                bad.append(name + " (not committed)")
            elif name.endswith(".json"):
                if json.load(open(a)) != json.load(open(b)):
                    bad.append(name)
            elif name.endswith(".npz"):
                za, zb = np.load(a), np.load(b)
                if set(za.files) != set(zb.files) or any(not np.array_equal(za[k], zb[k]) for k in za.files):
                    bad.append(name)
            elif not filecmp.cmp(a, b, shallow=False):
                bad.append(name)
        committed = {n for n in os.listdir(HERE) if n.endswith((".npy", ".npz", ".json"))}
        bad += sorted(n + " (no longer generated)" for n in committed - set(os.listdir(tmp)))
    return bad


if __name__ == "__main__":
    if "--check" in sys.argv[1:]:
        diff = check()
        print("golden files differ from a fresh generation:" if diff else "golden files reproduce", *diff)
        sys.exit(1 if diff else 0)
    main()
