"""spill instructions of the fused kernel, inside its main loop, per barrier-delimited segment (tools only):
hipcc ... -S --cuda-device-only wfft_test.hip -o x.s; python3 spills.py x.s"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
starts = [i for i, l in enumerate(lines) if re.match(r'^_ZN2ta11k_wfused_bp\S+:', l)]
for st in starts:
    name = lines[st].split(':')[0]
    end = next(i for i in range(st, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[st:end]
    # the outermost loop: from its header label to the last branch back to it
    hdr = next((i for i, l in enumerate(body) if 'Loop Header: Depth=1' in l), None)
    lo, hi = 0, len(body)
    if hdr is not None:
        label = body[hdr].split(':')[0].strip()
        backs = [i for i, l in enumerate(body) if re.search(r's_c?branch\S*\s+' + re.escape(label) + r'\b', l) and i > hdr]
        lo, hi = hdr, (backs[-1] if backs else len(body))
    seg, cnt, pre = 0, {}, [0, 0]
    for i, l in enumerate(body):
        m = re.search(r'scratch_(store|load)', l)
        if not (lo <= i <= hi):
            if m:
                pre[0 if m.group(1) == 'store' else 1] += 1
            continue
        if 's_barrier' in l:
            seg += 1
        if m:
            cnt[(seg, m.group(1))] = cnt.get((seg, m.group(1)), 0) + 1
    tot = {k: sum(v for (s_, kk), v in cnt.items() if kk == k) for k in ('store', 'load')}
    print(name[-34:-20], 'IN LOOP: spill stores', tot['store'], 'loads', tot['load'], '| outside', pre, '|',
          ' '.join('s%d:%s%d' % (k[0], k[1][0], v) for k, v in sorted(cnt.items())))
