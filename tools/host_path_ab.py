#!/usr/bin/env python3
"""bench.py's host_path_by_particle (VelocityAutocorr(fft=True).run() through the class, 10000 x 50000 x 3 float32 frames,
by-particle array) with and without the commit worker's page-locking ahead of the frame loop ("lock_ahead", round 6)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from transport_analysis_amd import _lib

orig = _lib.Context.__init__
for ahead in (0, 1, 0, 1):
    def init(self, device=0, _a=ahead):
        orig(self, device)
        self.set_option("lock_ahead", _a)
    _lib.Context.__init__ = init
    r = bench.host_path_by_particle(0, 10000, 3)
    print("lock_ahead", ahead, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k.endswith("_s") or k == "value"}), flush=True)
