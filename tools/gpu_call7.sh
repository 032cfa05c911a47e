#!/bin/bash
OUT=gpurun_out/r05g
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc $?"; tail -15 $OUT/pytest.log
cp gpurun_out/parity_margins.json $OUT/ 2>/dev/null
