"""The NumPy oracle on every host core: one worker process per usable CPU, each running
``numpy_oracle.vacf_fft`` (the reference's per-atom control flow) on its own block of atoms of
the benchmark tensor, which it generates itself (``oracle.synth``).  TEST INFRASTRUCTURE:
used only by bench.py's cpu_baseline leg (the "best CPU" line next to the one-core port).

Workers are plain child interpreters (``python -m oracle.parallel ...``): nothing of the
parent's GPU state, no re-import of the parent's main module.
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np


def usable_cpus():
    """CPUs this process may actually run on: the affinity mask capped by the cgroup quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except Exception:
        pass
    return max(1, n)


def _worker(argv):
    from . import numpy_oracle as orc
    from . import synth

    seed, T, n_cols_total, a_lo, a_hi, D = (int(x) for x in argv[:6])
    out = argv[6]
    t0 = time.perf_counter()
    lag = np.zeros(T)
    if a_hi > a_lo:
        v = synth.synthetic_block(seed, T, n_cols_total, a_lo * D, a_hi * D).reshape(T, a_hi - a_lo, D)
        bp, _ = orc.vacf_fft(v)
        lag = bp.sum(axis=1)
    np.save(out, lag)
    print(json.dumps({"seconds": time.perf_counter() - t0}), flush=True)


def vacf_fft_all_cores(seed, T, n_cols_total, n_atoms, D, n_workers=None):
    """Lag sums over atoms [0, n_atoms) with n_workers processes started together.  Returns
    (lagsum, seconds, n_workers): `seconds` is the longest worker's own time for generating
    and correlating its block (interpreter start-up excluded)."""
    n = n_workers or usable_cpus()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    edges = [n_atoms * i // n for i in range(n + 1)]
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for i in range(n):
            cmd = [sys.executable, "-m", "oracle.parallel", str(seed), str(T), str(n_cols_total), str(edges[i]),
                   str(edges[i + 1]), str(D), os.path.join(tmp, f"w{i}.npy")]
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            procs.append(subprocess.Popen(cmd, cwd=root, env=env, stdout=subprocess.PIPE, text=True))
        secs = []
        for p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError("oracle worker failed")
            secs.append(json.loads(out.strip().splitlines()[-1])["seconds"])
        lag = np.zeros(T)
        for i in range(n):
            lag += np.load(os.path.join(tmp, f"w{i}.npy"))
    return lag, max(secs), n


if __name__ == "__main__":
    _worker(sys.argv[1:])
