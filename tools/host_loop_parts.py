#!/usr/bin/env python3
"""Where a frame of the drop-in class's loop goes (10000 x 50000 x 3 float32): the stand-in trajectory's own
per-frame work, the binding's argument handling, and the native copy at several thread counts."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transport_analysis_amd import _lib  # noqa: E402
from transport_analysis_amd._base import native_rows, stage_frame_native  # noqa: E402
from transport_analysis_amd._mini_mda import ArrayUniverse  # noqa: E402

T, A = 4000, 50000
rng = np.random.default_rng(1)
blk = rng.standard_normal((250, A, 3), dtype=np.float32)
vel = np.empty((T, A, 3), dtype=np.float32)
for t in range(0, T, 250):
    vel[t:t + 250] = blk
u = ArrayUniverse(velocities=vel, positions=None)
traj = u.trajectory
ctx = _lib.Context(0)
(slab,) = ctx.stage_alloc(T, A, 3, n_slabs=1, dtype=np.float32)
rows = native_rows(u.atoms)
cols = [0, 1, 2]


def timed(name, fn, reps=2):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for i in range(T):
            fn(i)
        best = min(best, time.perf_counter() - t0)
    print(f"{name:58s} {best / T * 1e6:7.1f} us per frame")


timed("trajectory[i] (stand-in Timestep)", lambda i: traj[i])
timed("+ ts.velocities", lambda i: traj[i].velocities)
timed("+ frame_source(arr)", lambda i: _lib.frame_source(traj[i].velocities))
timed("+ ta_stage_frame (stage_frame_native)", lambda i: stage_frame_native(ctx, 0, i, traj[i], "velocities", cols, rows))
timed("NumPy: slab[i] = np.asarray(ag.velocities)[:, 0:3]", lambda i: slab[i].__setitem__(slice(None), np.asarray((traj[i], u.atoms.velocities)[1])[:, 0:3]))
src = _lib.frame_source(vel[7])
timed("ta_stage_frame alone, same source frame every time", lambda i: ctx.stage_frame(0, i, src, cols, rows[0]))
timed("np.copyto(slab[i], vel[i])", lambda i: np.copyto(slab[i], vel[i]))
print("threads:", _lib.lib().ta_stage_threads())
