#!/bin/bash
OUT=gpurun_out/r05f
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gemm16_gpu.py -m gpu -q -x > $OUT/pytest_g16.log 2>&1; echo "g16 rc $?"; tail -5 $OUT/pytest_g16.log
timeout 200 python tools/gemm16_probe.py > $OUT/gemm16_probe.txt 2>&1; grep planned $OUT/gemm16_probe.txt
timeout 400 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "bf16 or c5 or xcd or many_stream" > $OUT/pytest_bf16.log 2>&1; echo "bf16 rc $?"; tail -5 $OUT/pytest_bf16.log
timeout 100 python bench.py --config c5 --no-cpu-baseline 2>$OUT/bench_c5.err | tail -1 > $OUT/bench_c5.json
python - <<PY
import json
d = json.load(open("$OUT/bench_c5.json"))
print("c5", d["value"], d["ms_per_step"], d["multi_gpu_shard_path"])
for k, v in sorted(d["kernels"].items()):
    print("  %-30s %.1f x %.1f" % (k, v["avg_us"], v["launches_per_step"]))
PY
