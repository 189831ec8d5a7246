#!/bin/bash
# the headline path at equal data volume (~24 GB) over trajectory lengths: one line per length
# (n_frames, n_atoms, M, ms/step, lag-points/s, fraction of the HBM roof)
for cfg in "600 1600000" "1000 1000000" "1500 650000" "2000 500000" "2500 400000" "3000 330000" "3500 285000" "4000 250000" "4600 217000" "5000 200000" "6000 166000" "7000 142000" "8000 125000" "9000 111000" "10000 100000"; do set -- $cfg; python bench.py --frames $1 --atoms $2 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-path --no-check --no-kernel-split 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print(d['config']['n_frames'], d['config']['n_atoms_total'], d['config']['fft_plan']['M'], round(d['ms_per_step'],3), '%.3e'%d['value'], round(d['roofline']['frac'],3))
"; done
