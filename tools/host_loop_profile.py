#!/usr/bin/env python3
"""The class's frame loop at bench.py's host_path_by_particle shape, with the time spent inside
ta_stage_frame and ta_stage_commit summed separately (by_particle on and off: the page-locking helper thread)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transport_analysis_amd import VelocityAutocorr, _lib  # noqa: E402
from transport_analysis_amd._mini_mda import ArrayUniverse  # noqa: E402

T, A = 10000, 50000
rng = np.random.default_rng(11)
blk = rng.standard_normal((250, A, 3), dtype=np.float32)
vel = np.empty((T, A, 3), dtype=np.float32)
for t in range(0, T, 250):
    vel[t:t + 250] = blk
u = ArrayUniverse(velocities=vel, positions=None)
acc = {"frame": 0.0, "commit": 0.0}
orig_frame, orig_commit = _lib.Context.stage_frame, _lib.Context.stage_commit


def t_frame(self, *a):
    t0 = time.perf_counter()
    orig_frame(self, *a)
    acc["frame"] += time.perf_counter() - t0


def t_commit(self, *a):
    t0 = time.perf_counter()
    orig_commit(self, *a)
    acc["commit"] += time.perf_counter() - t0


_lib.Context.stage_frame, _lib.Context.stage_commit = t_frame, t_commit
for byp in (True, False, True):
    acc["frame"] = acc["commit"] = 0.0
    an = VelocityAutocorr(u.atoms, fft=True, by_particle=byp)
    marks = {}
    oc = an._conclude

    def tc():
        marks["loop_end"] = time.perf_counter()
        oc()

    an._conclude = tc
    op = an._prepare

    def tp():
        a0 = time.perf_counter()
        op()
        marks["prepare"] = time.perf_counter() - a0

    an._prepare = tp
    t0 = time.perf_counter()
    an.run()
    print(f"by_particle={byp}: run() up to _conclude {marks['loop_end'] - t0:.3f} s, of which _prepare {marks['prepare']:.3f} s, ta_stage_frame {acc['frame']:.3f} s, "
          f"ta_stage_commit {acc['commit']:.3f} s (first commit included)")
    del an
