#!/bin/bash
# The kernel-trace pass of tools/profile.sh alone, for several bench lines in one GPU-box call (after a change that moves kernel
# durations but not bytes): bash tools/profile_trace.sh r04 "" r04s8 "--streams-per-gpu 8" r04c5 "--config c5"
# -> gpurun_out/prof_<tag>/summary/<tag>_rocprofv3_kernel_stats.csv (+ domain stats)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
while [ $# -ge 2 ]; do
  TAG=$1; EXTRA=$2; shift 2
  OUT=gpurun_out/prof_$TAG
  CMD="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0 --launch eager $EXTRA"
  mkdir -p $OUT
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
  python tools/profile_summary.py $OUT $TAG "$CMD" | head -1
done
