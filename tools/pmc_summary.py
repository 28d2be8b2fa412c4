"""Summarise the rocprofv3 PMC passes written by collect_profiles.sh.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch (summed over the XCDs).  The
calibration kernels (tools/pmc_calib.hip) have exactly known traffic, so the ratio printed for them is
the correction to apply on this chip/driver: on gfx950 + ROCm 7.2 both came out at 1.00 for the access
shapes the engine uses (16 B/lane gathers of 64-byte blocks, 8 B/lane coalesced stores), i.e. NO
extra factor is applied; single-byte strided stores are the known outlier (6.3x write amplification).
"""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]


def per_kernel(dirname):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return agg


res = {"unit": "bytes per launch (mean over launches)", "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in per_kernel(os.path.join(out, "pmc_" + c)).items():
        if "bg" not in k and "fm_backward" not in k:
            continue
        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        res["kernels"].setdefault(short, {})[c] = {"launches": len(v), "mean_bytes": sum(v) / len(v)}
# the workload the counters were collected on (bench.py only reports `traffic` for the same launch shape)
for ln in open(os.path.join(out, "pmc_FETCH_SIZE.log")):
    if ln.startswith("{") and '"metric"' in ln:
        d = json.loads(ln)
        res["launch_shape"] = {"sw_pairs_per_launch": d["roofline"].get("pairs_per_launch"),
                               "fm_queries_per_launch": ((d.get("fm") or {}).get("roofline") or {}).get("queries_per_launch"),
                               "banded_pairs_per_launch": (d.get("banded") or {}).get("pairs_per_launch")}
json.dump(res, open(os.path.join(out, tag + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))

lines = []
expect = {  # kernel name prefix -> (counter, expected bytes) as printed by pmc_calib
}
for f in glob.glob(os.path.join(out, "cal_*.log")):
    for ln in open(f):
        if ln.startswith("EXPECT"):
            _, name, counter, nbytes = ln.split()
            expect[(name, counter)] = float(nbytes)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in per_kernel(os.path.join(out, "cal_" + c)).items():
        for (name, counter), nbytes in expect.items():
            if counter == c and name in k:
                mean = sum(v) / len(v)
                lines.append(f"{name:24s} {c:10s} measured {mean/1e9:9.4f} GB  expected {nbytes/1e9:9.4f} GB  ratio {mean/nbytes:6.3f}")
if lines:
    open(os.path.join(out, tag + "_pmc_calibration.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
