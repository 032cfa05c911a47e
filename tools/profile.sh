#!/bin/bash
# Profiling recipe behind profiles/rNN_* (run on the GPU box: gpurun -- 'bash tools/profile.sh r03').
# Four separate passes of the SAME command (MI355X_MICROARCH.md HBM/rocprofv3 section: counters never share a run
# with tracing), every pass under `timeout` (a rocprofv3 run that never exits would burn the box's budget):
#   1. --kernel-trace --stats                                   -> per-kernel average durations
#   2. --pmc FETCH_SIZE                                         -> HBM/MALL read traffic per launch
#   3. --pmc WRITE_SIZE                                         -> write traffic per launch
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE  -> MFMA pipe utilisation per kernel
TAG=${1:-r06}
shift
EXTRA="$@"
OUT=gpurun_out/prof_$TAG
CMD="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0 --launch eager $EXTRA"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- $CMD > $OUT/pmc_mfma.log 2>&1
python tools/profile_summary.py $OUT $TAG "$CMD"
