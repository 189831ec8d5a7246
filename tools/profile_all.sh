#!/bin/bash
# Run on the GPU box (via gpurun): the round's rocprofv3 evidence for every bench workload.
#   1. headline (configs[2] tensor, lag sums): --kernel-trace --stats over >= 10 timed calls after
#      >= 3 warm-ups, next to the hipEvent medians of the SAME run; FETCH/WRITE/TCC passes
#   2. HBM-traffic passes (FETCH_SIZE, WRITE_SIZE in separate runs) of the other workloads:
#      by-particle, direct configs[3] (matrix cores, with and without the by-particle array), Helfand float64 and float32
#      shares (time-packed matrix-core kernels, with and without the by-particle array), helfand_fft, 20000-frame path,
#      32 frames x 15.6 M atoms with the by-particle array (k_short)
# Everything lands in gpurun_out/prof_<tag>*/; hbm_traffic.json accumulates the entries, keyed by
# the library's hash.  usage: profile_all.sh TAG [quick]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05}
QUICK=${2:-}
COMMON="--no-cpu-baseline --no-other-configs --no-host-path --no-check --no-kernel-split --no-clock-probe"
export TA_TRAFFIC_MERGE=$R/gpurun_out/hbm_traffic_$TAG.json
rm -f $TA_TRAFFIC_MERGE
run() {  # name key kernels steps_total args...
  local name=$1 key=$2 kern=$3 nst=$4; shift 4
  TA_TRAFFIC_KEY=$key TA_TRAFFIC_KERNEL=$kern TA_PROFILE_STEPS=$nst bash $R/tools/profile_bench.sh ${TAG}_$name "$@" $COMMON > $R/gpurun_out/prof_${TAG}_$name.log 2>&1
  [ -f $R/gpurun_out/prof_${TAG}_$name/hbm_traffic.json ] && cp $R/gpurun_out/prof_${TAG}_$name/hbm_traffic.json $TA_TRAFFIC_MERGE
  echo "== $name"; tail -3 $R/gpurun_out/prof_${TAG}_$name.log
}
run c3 fft_10000x100000x3 k_wsplit_accum 13 --steps 10 --warmup 3
[ "$QUICK" = quick ] && exit 0
run c3bp fft_10000x100000x3_bp k_wsplit_accum+k_winverse+k_bp_transpose+k_sum_partials 3 --steps 2 --warmup 1 --by-particle
run direct direct_5000x50000x3 k_band_bp_vacf+k_bandbp_gather 3 --steps 2 --warmup 1 --mode direct --frames 5000 --atoms 50000
run directbp direct_5000x50000x3_bp k_band_bp_vacf+k_bp_transpose+k_sum_partials 3 --steps 2 --warmup 1 --mode direct --frames 5000 --atoms 50000 --by-particle
run helf64 helfand_20000x25000x3 k_helfand_product+k_band_bp_helf+k_bandbp_gather 2 --steps 1 --warmup 1 --mode helfand --frames 20000 --atoms 25000
run helf64bp helfand_20000x25000x3_bp k_helfand_product+k_band_bp_helf+k_bp_transpose+k_sum_partials 2 --steps 1 --warmup 1 --mode helfand --frames 20000 --atoms 25000 --by-particle
run helf32 helfand_20000x25000x3_f32 k_helfand_product32+k_band32_tp+k_bandbp_gather 2 --steps 1 --warmup 1 --mode helfand --float32 --frames 20000 --atoms 25000
run helf32bp helfand_20000x25000x3_bp_f32 k_helfand_product32+k_band32_tp+k_bp_transpose+k_sum_partials 2 --steps 1 --warmup 1 --mode helfand --float32 --frames 20000 --atoms 25000 --by-particle
run hfft helfand_20000x25000x3_hfft k_helfand_product+k_wsplit_accum+k_sum_partials+k_winverse+k_helfand_combine 3 --steps 2 --warmup 1 --mode helfand --helfand-fft --frames 20000 --atoms 25000
run long fft_20000x25000x3 k_wsplit_accum 4 --steps 3 --warmup 1 --frames 20000 --atoms 25000
run longbp fft_20000x25000x3_bp k_wsplit_accum+k_winverse+k_bp_transpose+k_sum_partials 3 --steps 2 --warmup 1 --frames 20000 --atoms 25000 --by-particle
run slab32 fft_10000x100000x3_slab32 k_wsplit_accum 13 --steps 10 --warmup 3 --slab32
run slab32bp fft_10000x100000x3_bp_slab32 k_wsplit_accum+k_winverse+k_bp_transpose+k_sum_partials 3 --steps 2 --warmup 1 --slab32 --by-particle
run short fft_32x15624960x3_bp k_short+k_sum_partials 4 --steps 3 --warmup 1 --frames 32 --atoms 15624960 --by-particle
run shortls fft_32x15624960x3 k_short 7 --steps 5 --warmup 2 --frames 32 --atoms 15624960
cat $TA_TRAFFIC_MERGE
