#!/usr/bin/env python3
"""Frame loop of the drop-in class with and without the native per-frame fill (ta_stage_frame):
bench.py's host_path_by_particle (VelocityAutocorr(fft=True).run(), 10000 x 50000 x 3 float32 frames),
$TA_AMD_NATIVE_STAGING=1 / 0, and the helper-thread count.  -> profiles/r05_host_loop.txt"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

from transport_analysis_amd import _lib  # noqa: E402

print("# ta_stage_threads =", _lib.lib().ta_stage_threads(), " host cpus =", len(os.sched_getaffinity(0)))
for native in ("1", "0", "1"):
    os.environ["TA_AMD_NATIVE_STAGING"] = native
    r = bench.host_path_by_particle(0, 10000, 3)
    print("native staging", native, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "what"}))
