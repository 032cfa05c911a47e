#!/bin/bash
# One GPU-box call at the end of a change: the whole -m gpu suite, then the bench lines whose numbers the docs quote.
# Usage (through gpurun): bash tools/round_check.sh TAG   -> gpurun_out/check_TAG/{pytest.log, bench_*.json}
TAG=${1:-r04}
OUT=gpurun_out/check_$TAG
mkdir -p $OUT
timeout 420 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $OUT/pytest.log
timeout 100 python bench.py 2>$OUT/bench_n1.err | tail -1 > $OUT/bench_n1.json
timeout 100 python bench.py --config c3 --no-cpu-baseline 2>$OUT/bench_c3.err | tail -1 > $OUT/bench_c3.json
timeout 100 python bench.py --config c5 --no-cpu-baseline 2>$OUT/bench_c5.err | tail -1 > $OUT/bench_c5.json
python - <<PY
import json
for n in ("n1", "c3", "c5"):
    try:
        d = json.load(open("$OUT/bench_%s.json" % n))
        print(n, d["value"], d["ms_per_step"], {k: round(v["avg_us"], 1) for k, v in d.get("kernels", {}).items()})
    except Exception as ex:
        print(n, "no line:", ex)
PY
