#!/bin/bash
# The round's whole evidence in ONE GPU-box call (gpurun --timeout 2400 -- 'bash tools/profile_all.sh'): the four rocprofv3 passes of
# tools/profile.sh for the default line, 8 streams per GPU and --config c5, and --config c4, then tools/round_check.sh
# (the -m gpu suite + every bench line).  Summaries land in gpurun_out/profiles_new/ (copy them to profiles/), bench lines and parity
# margins in gpurun_out/check_r06/.
bash tools/profile.sh r06 2>&1 | tail -2
bash tools/profile.sh r06s8 --streams-per-gpu 8 2>&1 | tail -2
bash tools/profile.sh r06c5 --config c5 2>&1 | tail -2
bash tools/profile.sh r06c4 --config c4 2>&1 | tail -2
mkdir -p profiles_new && cp gpurun_out/prof_r06*/summary/* profiles_new/ 2>/dev/null; cp -r profiles_new gpurun_out/
cp gpurun_out/prof_r06*/summary/*pmc_traffic.json profiles/ 2>/dev/null
OUT=gpurun_out/check_r06
ROUND_CHECK_ALLOW_STALE_PMC=1 bash tools/round_check.sh r06
# the probes whose tables the docs quote
P=gpurun_out/profiles_new
timeout 300 python tools/scale_probe.py 2>/dev/null | grep -v amdgpu.ids > $P/r06_scale_probe.txt
timeout 300 python tools/scale_probe.py bf16 2>/dev/null | grep -v amdgpu.ids >> $P/r06_scale_probe.txt
timeout 120 python tools/tail_timing.py 80 2>/dev/null | grep -v amdgpu.ids > $P/r06_tail_ops.txt
timeout 200 python tools/gemm16_probe.py --copies 2>/dev/null | grep -v amdgpu.ids > $P/r06_gemm16_copies.txt
timeout 200 python tools/gemm16_anatomy.py 2>/dev/null | grep -v amdgpu.ids > $P/r06_gemm16_anatomy.txt
: > $P/r06_adapter_verify_ab.txt
for cfg in "4 400 50 1 0 1" "4 400 50 1 0 0" "4 400 50 1 1 1" "4 400 50 0 0 1" "8 400 50 1 0 1" "8 400 50 0 0 1"; do
  echo "adapter $cfg: $(LD_LIBRARY_PATH=kaldi-lstm_amd timeout 100 tools/kaldi_adapter_bench $cfg 2>&1 | cut -c1-330)" >> $P/r06_adapter_verify_ab.txt
done
cp $OUT/bench_*.json $OUT/parity_margins.json $P/ 2>/dev/null
