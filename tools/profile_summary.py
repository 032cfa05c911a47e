"""Collect the rocprofv3 outputs of tools/profile.sh into gpurun_out/prof_<tag>/summary/ (files named as they are
committed under profiles/): kernel stats CSV, the two PMC CSVs, and <tag>_pmc_traffic.json with HBM bytes per launch
= (2*FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE
reports half of wide coalesced reads)."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


ks = find("trace", "*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, f"{tag}_rocprofv3_kernel_stats.csv"))
ds = find("trace", "*domain_stats.csv")
if ds:
    shutil.copy(ds, os.path.join(dst, f"{tag}_rocprofv3_domain_stats.csv"))
agg = defaultdict(lambda: defaultdict(list))
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    shutil.copy(f, os.path.join(dst, f"{tag}_pmc_{ctr}_counter_collection.csv"))
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == ctr:
            name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("klstm::", "")
            name = re.sub(r"\(.*\)$", "", name)
            agg[name][ctr].append(float(row["Counter_Value"]))
kern = {}
for name, c in agg.items():
    fe = sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"]))
    wr = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"]))
    kern[name] = {"fetch_kb_raw": fe, "write_kb_raw": wr, "hbm_bytes_per_launch": (2 * fe + wr) * 1024,
                  "launches": len(c["FETCH_SIZE"])}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile.sh) on "
                     "`python bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch graph`, S=4, T=20, 40/800/512",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE reports 1/2 of wide "
                         "coalesced reads; WRITE_SIZE uncalibrated)",
           "kernels": kern}, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print("summary in", dst, ":", sorted(os.listdir(dst)))
