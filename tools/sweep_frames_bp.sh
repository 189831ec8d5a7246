#!/bin/bash
# by-particle FFT path at equal data volume (24 GB in, 8 GB out) over trajectory lengths
for cfg in "300 3000000" "1000 1000000" "1500 650000" "2000 500000" "2500 400000" "4000 250000" "6000 166000" "10000 100000"; do set -- $cfg; python bench.py --by-particle --frames $1 --atoms $2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-path --no-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print(d['config']['n_frames'], d['config']['n_atoms_total'], d['config']['fft_plan']['M'], round(d['ms_per_step'],3), '%.3e'%d['value'], 'kernels', round(d['roofline']['kernel_ms'],2))
"; done
