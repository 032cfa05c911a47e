#!/bin/bash
OUT=gpurun_out/r05j
mkdir -p $OUT
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "fuse_update or bf16 or c5 or c4" > $OUT/pytest.log 2>&1; echo "rc $?"; tail -8 $OUT/pytest.log
timeout 100 python bench.py --config c5 --no-cpu-baseline 2>$OUT/bench_c5.err | tail -1 > $OUT/bench_c5.json
timeout 100 python bench.py --config c5 --no-cpu-baseline --option fuse_update=0 2>>$OUT/bench_c5.err | tail -1 > $OUT/bench_c5_nofuse.json
python - <<PY
import json
for n in ("bench_c5", "bench_c5_nofuse"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, round(d["value"]), round(d["ms_per_step"], 4), d["multi_gpu_shard_path"]["ms_per_step"])
    print("   ", {k: round(v["avg_us"], 1) for k, v in d["kernels"].items() if "layer1" in k})
PY
