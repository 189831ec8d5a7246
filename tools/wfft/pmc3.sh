#!/bin/bash
# FP64 instruction mix of the harness' forward kernel: pmc3.sh TAG [BINARY-SUFFIX] [harness args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; shift
BIN=$R/tools/wfft/wfft_test; if [ -n "${1:-}" ] && [ -x "$R/tools/wfft/wfft_test_$1" ]; then BIN=$R/tools/wfft/wfft_test_$1; shift; fi
OUT=$R/gpurun_out/pmc3_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS=${@:-"time 30000 10000 2 0"}
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/s1 -- $BIN $ARGS > $OUT/log1.txt 2>&1
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'accum' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
for k in sorted(m): print(f"$TAG {k:28s} {m[k]:.6g}")
f64=m.get('SQ_INSTS_VALU_ADD_F64',0)+m.get('SQ_INSTS_VALU_MUL_F64',0)+m.get('SQ_INSTS_VALU_FMA_F64',0)
if f64: print(f"$TAG FMA share of FP64 instructions {m['SQ_INSTS_VALU_FMA_F64']/f64:.3f}; FP64 share of vector instructions {f64/m['SQ_INSTS_VALU']:.3f}")
PY
find $OUT -name '*.csv' -size +1M -delete
