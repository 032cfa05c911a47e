#!/bin/bash
# The round's whole evidence in ONE GPU-box call (gpurun --timeout 2400 -- 'bash tools/profile_all.sh'): the four rocprofv3 passes of
# tools/profile.sh for the default line, 8 streams per GPU and --config c5, and --config c4, then tools/round_check.sh
# (the -m gpu suite + every bench line).  Summaries land in gpurun_out/profiles_new/ (copy them to profiles/), bench lines and parity
# margins in gpurun_out/check_r05/.
bash tools/profile.sh r05 2>&1 | tail -2
bash tools/profile.sh r05s8 --streams-per-gpu 8 2>&1 | tail -2
bash tools/profile.sh r05c5 --config c5 2>&1 | tail -2
bash tools/profile.sh r05c4 --config c4 2>&1 | tail -2
mkdir -p profiles_new && cp gpurun_out/prof_r05*/summary/* profiles_new/ 2>/dev/null; cp -r profiles_new gpurun_out/
cp gpurun_out/prof_r05*/summary/*pmc_traffic.json profiles/ 2>/dev/null
ROUND_CHECK_ALLOW_STALE_PMC=1 bash tools/round_check.sh r05
