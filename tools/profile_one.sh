#!/bin/bash
# one more case of tools/profile_all.sh for the SAME library, merged into gpurun_out/hbm_traffic_TAG.json:
#   profile_one.sh TAG name key kernels steps_total bench-args...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; name=$2; key=$3; kern=$4; nst=$5; shift 5
COMMON="--no-cpu-baseline --no-other-configs --no-host-path --no-check --no-kernel-split --no-clock-probe"
export TA_TRAFFIC_MERGE=$R/gpurun_out/hbm_traffic_$TAG.json
[ -f $TA_TRAFFIC_MERGE ] || cp $R/profiles/hbm_traffic.json $TA_TRAFFIC_MERGE
TA_TRAFFIC_KEY=$key TA_TRAFFIC_KERNEL=$kern TA_PROFILE_STEPS=$nst bash $R/tools/profile_bench.sh ${TAG}_$name "$@" $COMMON > $R/gpurun_out/prof_${TAG}_$name.log 2>&1
[ -f $R/gpurun_out/prof_${TAG}_$name/hbm_traffic.json ] && cp $R/gpurun_out/prof_${TAG}_$name/hbm_traffic.json $TA_TRAFFIC_MERGE
tail -4 $R/gpurun_out/prof_${TAG}_$name.log
