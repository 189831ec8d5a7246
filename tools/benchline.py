#!/usr/bin/env python3
"""Compact one-line summary of bench.py JSON lines read from stdin."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(sys.argv[1] if len(sys.argv) > 1 else "", d["config"]["workload"], "|", d["dtype"], "| kernel_ms",
          round(r["kernel_ms"], 3), "| frac", round(r["frac"], 4), "| value %.3e" % d["value"], "|", d["check"], flush=True)
