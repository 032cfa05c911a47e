#!/bin/bash
# One GPU-box call at the end of a change: the whole -m gpu suite, then the bench lines whose numbers the docs quote.
# Usage (through gpurun): bash tools/round_check.sh TAG   -> gpurun_out/check_TAG/{pytest.log, bench_*.json, parity_margins.json}
# Refuses when the committed PMC summary of the round (profiles/TAG_pmc_traffic.json) is older than the kernels it describes
# (the source hash recorded next to it): re-run tools/profile.sh TAG [--streams-per-gpu 8 | --config c5] first.
TAG=${1:-r06}
OUT=gpurun_out/check_$TAG
mkdir -p $OUT
python - <<PY || exit 3
import json, os, sys
sys.path.insert(0, ".")
import importlib.util
spec = importlib.util.spec_from_file_location("b", "kaldi-lstm_amd/build.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
want = m.source_hash()
stale = []
for suffix in ("", "s8", "c5"):
    f = "profiles/${TAG}%s_pmc_traffic.json" % suffix
    if not os.path.exists(f):
        stale.append(f + " (missing)"); continue
    got = json.load(open(f)).get("library_source_hash")
    if got != want:
        stale.append("%s (kernels %s, PMC pass %s)" % (f, want[:12], (got or "unrecorded")[:12]))
if stale and not os.environ.get("ROUND_CHECK_ALLOW_STALE_PMC"):
    print("round_check: the PMC summaries are older than csrc/: " + "; ".join(stale) + " -- run tools/profile.sh first (ROUND_CHECK_ALLOW_STALE_PMC=1 to go on anyway)")
    sys.exit(3)
PY
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $OUT/pytest.log
cp gpurun_out/parity_margins.json $OUT/ 2>/dev/null
timeout 200 python bench.py 2>$OUT/bench_n1.err | tail -1 > $OUT/bench_n1.json
timeout 100 python bench.py --config c1 2>$OUT/bench_c1.err | tail -1 > $OUT/bench_c1.json
timeout 100 python bench.py --config c3 2>$OUT/bench_c3.err | tail -1 > $OUT/bench_c3.json
timeout 100 python bench.py --config c4 --no-cpu-baseline 2>$OUT/bench_c4.err | tail -1 > $OUT/bench_c4.json
timeout 100 python bench.py --config c5 --no-cpu-baseline 2>$OUT/bench_c5.err | tail -1 > $OUT/bench_c5.json
python - <<PY
import json
for n in ("n1", "c1", "c3", "c4", "c5"):
    try:
        d = json.load(open("$OUT/bench_%s.json" % n))
        print(n, round(d["value"]), round(d["ms_per_step"], 4), {k: round(v["avg_us"], 1) for k, v in d.get("kernels", {}).items()})
        if n == "n1":
            for k in ("c4", "c5", "kaldi_adapter", "s8_per_gpu", "strict_f32", "ragged"):
                v = d.get(k) or {}
                print("   ", k, round(v.get("value", 0)), v.get("ms_per_step"))
    except Exception as ex:
        print(n, "no line:", ex)
PY
