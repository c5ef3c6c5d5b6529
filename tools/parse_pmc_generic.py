"""rocprofv3 counter_collection.csv -> per (kernel, grid) sums of every counter, plus ratios used in DESIGN section 3."""
import collections
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if len(sys.argv) > 3 and sys.argv[3] not in k:
        continue
    k = k + " grid=" + r.get("Grid_Size", "")
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
out = {}
for k, v in per.items():
    d = {n: v[n] / max(cnt[(k, n)], 1) for n in v}       # per dispatch
    d["dispatches"] = max(cnt[(k, n)] for n in v)
    act = d.get("GRBM_GUI_ACTIVE")
    if act and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["mfma_pipe_util"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (act / 8 * 1024), 4)
    if "SQ_WAVE_CYCLES" in d:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            if n in d:
                d[n + "/WAVE_CYCLES"] = round(d[n] / d["SQ_WAVE_CYCLES"], 4)
    out[k] = d
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k in sorted(out):
    print(k[:80], {a: (round(b, 4) if isinstance(b, float) and b < 10 else int(b)) for a, b in out[k].items()})
