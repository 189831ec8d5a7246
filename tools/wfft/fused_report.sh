#!/bin/bash
# FFT VACF by particle, the library's chain (k_wsplit_accum by particle -> spectra in HBM -> k_winverse)
# against the ONE-kernel form of tools/wfft/wfused.hpp (all four passes of an atom from one read of its rows, the power
# spectrum and the inverse transform on the compute unit), same box, 10000 frames x 100000 atoms x 3:
#   library          tools/timeline_case.py fft 10000 100000 --bp [--f32]   (kernel timeline from the library's events)
#   wfft_test_fu8    the fused kernel on 8 waves x 256 registers (two per SIMD)
#   wfft_test_fu4    ... on 4 waves x 512 registers (-DWF_NW_R0=20 -DWF_NW_VAL=4)
# per run: rocm-smi package power and sclk sampled while it loops, hipEvent time, in-kernel cycles per phase (wave 0).
# Run on the GPU box: tools/wfft/fused_report.sh -> gpurun_out/r06_bp_fused_raw.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/r06_bp_fused_raw.txt}
mkdir -p $(dirname $OUT); cd /tmp
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/^GPU\[0\]\s*: //; s/=\{10,\}//g' | tr '\n' ';'; echo; }
sample() { while kill -0 $1 2>/dev/null; do smi; sleep 0.4; done | tail -5 | head -4; wait $1; }
{
echo "# fused_report.sh $(date -u +%FT%TZ)"
for f32 in "" "--f32"; do
  echo "## library chain, slab ${f32:-float64}"
  python3 $R/tools/timeline_case.py fft 10000 100000 --bp $f32 --reps 300 > /tmp/fr.log 2>&1 &
  sample $!
  tail -2 /tmp/fr.log
done
for v in fu8 fu4; do
  for f32 in 0 1; do
    BIN=$R/tools/wfft/wfft_test_$v
    echo "## wfft_test_$v  float32 rows: $f32  (sha $(sha256sum $BIN | cut -c1-16))"
    WF_R0=20 timeout -k 10 100 $BIN fcheck 0 $f32 2>&1 | tail -1
    WF_R0=20 timeout -k 10 200 $BIN ftime 100000 10000 120 3 $f32 > /tmp/fr.log 2>&1 &
    sample $!
    cat /tmp/fr.log
  done
done
} 2>&1 | tee $OUT
