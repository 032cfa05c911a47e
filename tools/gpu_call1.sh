#!/bin/bash
# first GPU call of round 5: whole -m gpu suite (no -x: every failing bar is wanted), margins, the default bench line, the adapter tool
OUT=gpurun_out/r05a
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc $?"; tail -15 $OUT/pytest.log
cp gpurun_out/parity_margins.json $OUT/ 2>/dev/null
timeout 200 python bench.py 2>$OUT/bench_n1.err | tail -1 > $OUT/bench_n1.json
python - <<PY
import json
d = json.load(open("$OUT/bench_n1.json"))
print("n1", d["value"], d["ms_per_step"])
for k in ("c4", "c5", "kaldi_adapter", "s8_per_gpu", "strict_f32"):
    print(k, d.get(k))
PY
tail -3 $OUT/bench_n1.err
