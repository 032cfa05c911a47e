#!/bin/bash
OUT=gpurun_out/r05d
mkdir -p $OUT
timeout 200 python -m pytest tests/test_gemm16_gpu.py -m gpu -q -x -k "not engine" > $OUT/pytest_g16.log 2>&1; echo "g16 rc $?"; tail -3 $OUT/pytest_g16.log
for cfg in "1 3" "0 3" "1 4" "0 4"; do
  set -- $cfg
  KLSTM_G16_ROT=$1 KLSTM_G16_NBUF=$2 timeout 200 python tools/gemm16_probe.py > $OUT/gemm16_probe_rot$1_nbuf$2.txt 2>&1; cat $OUT/gemm16_probe_rot$1_nbuf$2.txt
done
