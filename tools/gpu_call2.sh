#!/bin/bash
OUT=gpurun_out/r05b
mkdir -p $OUT
timeout 600 python -m pytest tests/test_persist_robustness_gpu.py tests/test_component.py tests/test_nnet.py -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc $?"; tail -8 $OUT/pytest.log
for args in "4 400 50 1 0 1" "4 400 50 1 0 0" "4 400 50 1 1 1" "4 400 50 1 1 0" "4 400 50 0 0 1" "8 400 50 1 0 1" "8 400 50 1 0 0" "8 400 50 0 0 1"; do
  echo "adapter $args: $(./tools/kaldi_adapter_bench $args | cut -c1-60)"
done | tee $OUT/adapter.txt
