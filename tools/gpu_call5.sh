#!/bin/bash
OUT=gpurun_out/r05e
mkdir -p $OUT
timeout 200 python -m pytest tests/test_gemm16_gpu.py -m gpu -q -x -k "not engine" > $OUT/pytest_g16.log 2>&1; echo "g16 rc $?"; tail -5 $OUT/pytest_g16.log
timeout 100 python tools/gemm16_anatomy.py 2>&1 | tee $OUT/gemm16_anatomy.txt
timeout 200 python tools/gemm16_probe.py 2>&1 | tee $OUT/gemm16_probe_nf2.txt

