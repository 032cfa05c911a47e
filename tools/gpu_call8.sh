#!/bin/bash
OUT=gpurun_out/r05h
mkdir -p $OUT
timeout 600 python -m pytest tests/test_range_gpu.py tests/test_gemm16_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc $?"; tail -6 $OUT/pytest.log
timeout 400 python -m pytest tests/test_engine_gpu.py -m gpu -q -k "fp16_products or fold" > $OUT/pytest2.log 2>&1
echo "pytest2 rc $?"; tail -3 $OUT/pytest2.log
bash tools/profile.sh r05c5 --config c5 2>&1 | tail -3
