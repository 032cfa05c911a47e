"""Collect the rocprofv3 outputs of tools/profile.sh into gpurun_out/prof_<tag>/summary/ (files named as they are
committed under profiles/): kernel stats CSV, the PMC CSVs, and <tag>_pmc_traffic.json with
  * HBM bytes per launch per kernel = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are
    in KB; on gfx950 FETCH_SIZE reports half of wide coalesced reads),
  * hbm_bytes_per_minibatch = sum over the dispatches that recur every minibatch / number of minibatches (= launches of k_grads:
    exactly one per minibatch) -- bench.py divides it by SURVEY 8(d)'s algorithmic bytes for `roofline.traffic_ratio`; what runs once
    per process (set_params' packing, zero-fills) is `hbm_bytes_setup_once`; `..._incl_setup` = everything / minibatches (round <= 3),
  * mfma_busy_frac per kernel = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs): the gfx94x MfmaUtil formula
    reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum) / (reduce(GRBM_GUI_ACTIVE,max) * SIMD_NUM) (counter_defs.yaml; ROCm 7.2 has no gfx950
    derived-counter section).  rocprofv3's CSV carries the SUM of GRBM_GUI_ACTIVE over its 8 XCC instances, which are all
    active for the whole dispatch, so max = sum/8.  Cross-check: the fold product's 2.62 GFLOP are 1.28 M 16x16x4 MFMAs of
    32 cycles = 41 M busy cycles, the counter reads 42.6 M."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def clean(name):
    name = re.sub(r"^void ", "", name).replace("klstm::", "")
    return re.sub(r"\(.*\)$", "", name)


ks = find("trace", "*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, f"{tag}_rocprofv3_kernel_stats.csv"))
ds = find("trace", "*domain_stats.csv")
if ds:
    shutil.copy(ds, os.path.join(dst, f"{tag}_rocprofv3_domain_stats.csv"))
agg = defaultdict(lambda: defaultdict(list))
for sub, ctrs in (("pmc_fetch", ("FETCH_SIZE",)), ("pmc_write", ("WRITE_SIZE",)),
                  ("pmc_mfma", ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    shutil.copy(f, os.path.join(dst, f"{tag}_{sub}_counter_collection.csv"))
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] in ctrs:
            agg[clean(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
kern = {}
total_bytes = 0.0
for name, c in agg.items():
    nf, nw = max(1, len(c["FETCH_SIZE"])), max(1, len(c["WRITE_SIZE"]))
    fe, wr = sum(c["FETCH_SIZE"]) / nf, sum(c["WRITE_SIZE"]) / nw
    kern[name] = {"fetch_kb_raw": fe, "write_kb_raw": wr, "hbm_bytes_per_launch": (2 * fe + wr) * 1024,
                  "launches": len(c["FETCH_SIZE"])}
    total_bytes += (2 * sum(c["FETCH_SIZE"]) + sum(c["WRITE_SIZE"])) * 1024      # both passes run the same dispatches
    if c["SQ_VALU_MFMA_BUSY_CYCLES"] and c["GRBM_GUI_ACTIVE"]:
        kern[name]["mfma_busy_frac"] = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(c["GRBM_GUI_ACTIVE"]) / 8 * 1024)
        kern[name]["sq_busy_cycles_per_launch"] = sum(c["SQ_BUSY_CYCLES"]) / max(1, len(c["SQ_BUSY_CYCLES"]))
# one gradient launch per LSTM layer and minibatch: k_grads (fp32; bench.py default / --config c4: 2 layers) or k_grads_bf16 (--config c5: 3 layers)
mcfg = re.search(r"--config[ =](c\d)", cmd)
layers = {"c4": 2, "c5": 3}.get(mcfg.group(1) if mcfg else "", 1)
nmb = max(kern.get("k_grads", {}).get("launches", 0),
          sum(v.get("launches", 0) for n, v in kern.items() if n.startswith("k_grads_bf16"))) // layers   # (k_grads_bf16<false | true>)
# steady state = the kernels that run once (or more) per minibatch; what runs once per PROCESS (set_params: k_update_repack_v, k_pack,
# k_split3, the zero-fills of the planes) is listed as setup and not charged to a minibatch
steady = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in kern.values() if nmb and v["launches"] >= nmb) / nmb if nmb else None
setup = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in kern.values() if not nmb or v["launches"] < nmb)
m = re.search(r"--streams-per-gpu[ =](\d+)", cmd)
try:        # which kernels these passes saw: the content hash of the library's sources (tools/round_check.sh refuses a stale summary)
    import importlib.util
    _spec = importlib.util.spec_from_file_location("klstm_build", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaldi-lstm_amd", "build.py"))
    _mod = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_mod)
    src_hash = _mod.source_hash()
except Exception:
    src_hash = None
doc = {"library_source_hash": src_hash,
       "source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; separate "
                 "runs, tools/profile.sh) on `" + cmd + "`, T=20, 40/800/512",
       "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads; "
                     "WRITE_SIZE uncalibrated)",
       "streams_per_gpu": int(m.group(1)) if m else {"c5": 32}.get(mcfg.group(1) if mcfg else "", 4),
       "config": mcfg.group(1) if mcfg else ("c3" if m and m.group(1) == "8" else "c2"), "lstm_layers": layers,
       "chain": "per-XCD persistent" if any(n.startswith("k_bwd_persist_xl") for n in kern) else
                "persistent" if any(n.startswith("k_bwd_persist") for n in kern) else "launches",   # (which kind of kernel runs BPTT, the dominant chain)
       "minibatches": nmb, "hbm_bytes_per_minibatch": steady, "hbm_bytes_setup_once": setup,
       "hbm_bytes_per_minibatch_incl_setup": total_bytes / nmb if nmb else None,
       "kernels": kern}
json.dump(doc, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print("summary in", dst, ":", sorted(os.listdir(dst)))
print("hbm bytes per minibatch:", doc["hbm_bytes_per_minibatch"])
