"""Summarise the rocprofv3 PMC passes written by collect_profiles.sh.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch (summed over the XCDs).  The
calibration kernels (tools/pmc_calib.hip) have exactly known traffic, so the ratio printed for them is
the correction to apply on this chip/driver: on gfx950 + ROCm 7.2 both came out at 1.00 for the access
shapes the engine uses (16 B/lane gathers of 64-byte blocks, 8 B/lane coalesced stores), i.e. NO
extra factor is applied; single-byte strided stores are the known outlier (6.3x write amplification).
The SQ_* counters are plain event counts per dispatch (wave instructions, cycles).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_sha  # noqa: E402

out, tag = sys.argv[1], sys.argv[2]
SHA = csrc_sha()  # the kernel sources these counters were collected with: bench.py reports them only while they match
OURS = ("bgsw", "bgband", "bgfm", "fm_backward", "fm_search_fast", "fq_", "se_", "sa_", "fmd_", "cigar_kernel", "pretty_kernel", "interval_rows", "pack2_", "ops_compact")


def short(k):
    return k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def per_kernel(dirname):
    """{counter: {kernel: [value per dispatch]}}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if not any(t in r["Kernel_Name"] for t in OURS):
                continue
            agg[r["Counter_Name"]][short(r["Kernel_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {c: {k: list(v.values()) for k, v in ks.items()} for c, ks in agg.items()}


def launch_shape(logfile):
    for ln in open(logfile):
        if ln.startswith("{") and '"metric"' in ln:
            d = json.loads(ln)
            k1 = d.get("k1_int32") or {}
            return {"sw_pairs_per_launch": d["roofline"].get("pairs_per_launch"),
                    "fm_queries_per_launch": ((d.get("fm") or {}).get("roofline") or {}).get("queries_per_launch"),
                    "banded_pairs_per_launch": (d.get("banded") or {}).get("pairs_per_launch"),
                    "k1_pairs_per_launch": ((next(iter(k1.values()), {})).get("roofline") or {}).get("pairs_per_launch"),
                    "sg_pairs_per_launch": ((d.get("semiglobal") or {}).get("roofline") or {}).get("pairs_per_launch"),
                    "ingest_bytes": int(((d.get("ingest") or {}).get("config") or {}).get("workload", "0 (0 bytes)").split("(")[-1].split()[0]),
                    "command": "bench.py --skip-cpu --skip-pipeline --skip-packed --fm-big-genome 0 --banded-pairs 98304 --steps 2 --warmup 0"}
    return {}


passes = sorted(glob.glob(os.path.join(out, "pmc_[0-9]")))
counters = {}
for p in passes:
    counters.update(per_kernel(p))
shape = launch_shape(os.path.join(out, "pmc_1.log")) if passes else {}

# ---- HBM traffic
res = {"unit": "bytes per launch (mean over launches)", "csrc_sha": SHA, "kernels": {}, "launch_shape": shape}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in counters.get(c, {}).items():
        res["kernels"].setdefault(k, {})[c] = {"launches": len(v), "mean_bytes": sum(v) * 1024.0 / len(v)}
# FASTQ ingest is many short kernels per call: bytes of all of them per bg_fastq_parse_dev call (the leg makes 4 calls)
fq = sum(sum(v) * 1024.0 for c in ("FETCH_SIZE", "WRITE_SIZE") for k, v in counters.get(c, {}).items() if k.startswith("fq_"))
res["ingest_bytes_per_call"] = fq / 4.0
# the FM search kernel on the index beyond the Infinity Cache (its own passes)
big = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = [v for k, vs in per_kernel(os.path.join(out, "big_" + c)).get(c, {}).items() if "fm_search_fast_kernel" in k for v in vs]
    if vals:
        big[c] = {"launches": len(vals), "mean_bytes": sum(vals) * 1024.0 / len(vals)}
logf = os.path.join(out, "big_FETCH_SIZE.log")
if big and os.path.exists(logf):
    for ln in open(logf):
        if ln.startswith("{") and '"fm_big"' in ln:
            fb = json.loads(ln)["fm_big"]
            big["queries_per_launch"] = fb["roofline"]["queries_per_launch"]
            big["index_bytes"] = fb["config"]["index_bytes"]
            big["alg_bytes_per_launch"] = fb["roofline"]["alg_bytes_per_query"] * fb["roofline"]["queries_per_launch"]
    res["fm_big"] = big
# the seed-and-extend leg of the same passes: every kernel of bg_seed_extend_batch_dev, bytes per call
SE_KERNELS = ("se_", "sa_sampled_get", "sa_raw", "interval_rows", "fm_search_fast_kernel<true", "fm_backward_search_kernel<false, true",
              "semiglobal", "sw_fill_pk16", "sw_traceback_kernel")
se = {"kernels": {}}
tot_bytes = 0.0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, vs in per_kernel(os.path.join(out, "big_" + c)).get(c, {}).items():
        if not k.startswith(("at::", "void at::", "elementwise_kernel", "rocprim", "void rocprim")) and any(t in k for t in SE_KERNELS):
            se["kernels"].setdefault(k, {})[c] = {"launches": len(vs), "mean_bytes": sum(vs) * 1024.0 / len(vs), "total_bytes": sum(vs) * 1024.0}
            tot_bytes += sum(vs) * 1024.0
if se["kernels"] and os.path.exists(logf):
    for ln in open(logf):
        if ln.startswith("{") and '"seed_extend"' in ln:
            sl = json.loads(ln)["seed_extend"]
            calls = sl["roofline"].get("calls_in_run") or 1
            se["reads_per_call"] = int(sl["config"]["workload"].split()[0])
            se["genome"] = sl["config"]["genome"]
            se["calls_in_run"] = calls
            se["bytes_per_call"] = tot_bytes / calls
            se["note"] = ("FETCH + WRITE of every kernel of the pipeline over the run / calls; includes the 65 536-pair headline leg's "
                          "K1p / K2 launches of the same command (< 2 % of the pipeline's candidates)")
    if "bytes_per_call" in se:
        res["seed_extend"] = se
json.dump(res, open(os.path.join(out, tag + "_pmc_traffic.json"), "w"), indent=1)

# ---- issue counters
sq = {"unit": "events per launch (mean over launches)", "csrc_sha": SHA, "kernels": {}, "launch_shape": shape}
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
    for k, v in counters.get(c, {}).items():
        sq["kernels"].setdefault(k, {})[c] = sum(v) / len(v)
json.dump(sq, open(os.path.join(out, tag + "_sq_counters.json"), "w"), indent=1)

# ---- LDS bank conflicts
lds = {"unit": "events per launch (mean over launches)", "csrc_sha": SHA, "kernels": {}, "launch_shape": shape}
for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS"):
    for k, v in counters.get(c, {}).items():
        lds["kernels"].setdefault(k, {})[c] = sum(v) / len(v)
for k, c in lds["kernels"].items():
    a = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    c["conflict_over_active"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / a, 4) if a else None
json.dump(lds, open(os.path.join(out, tag + "_lds_counters.json"), "w"), indent=1)
with open(os.path.join(out, tag + "_lds_counters.txt"), "w") as f:
    f.write("# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS (one pass, kernel-trace only)\n")
    f.write("# of: " + shape.get("command", "?") + " ; per launch, in millions (cycles / instructions)\n")
    f.write("%-72s %14s %18s %21s %16s\n" % ("kernel", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "conflict/active"))
    for k, c in sorted(lds["kernels"].items()):
        f.write("%-72s %14.2f %18.2f %21.2f %16s\n" % (k[:72], c.get("SQ_INSTS_LDS", 0) / 1e6, c.get("SQ_LDS_IDX_ACTIVE", 0) / 1e6,
                                                  c.get("SQ_LDS_BANK_CONFLICT", 0) / 1e6,
                                                  "-" if c["conflict_over_active"] is None else "%.1f %%" % (100 * c["conflict_over_active"])))
print(json.dumps({"traffic_kernels": len(res["kernels"]), "sq_kernels": len(sq["kernels"]), "lds_kernels": len(lds["kernels"])}))

# ---- calibration
lines = []
expect = {}
for f in glob.glob(os.path.join(out, "cal_*.log")):
    for ln in open(f):
        if ln.startswith("EXPECT"):
            _, name, counter, nbytes = ln.split()
            expect[(name, counter)] = float(nbytes)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, "cal_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    for k, v in agg.items():
        for (name, counter), nbytes in expect.items():
            if counter == c and name in k:
                mean = sum(v) / len(v)
                lines.append(f"{name:24s} {c:10s} measured {mean/1e9:9.4f} GB  expected {nbytes/1e9:9.4f} GB  ratio {mean/nbytes:6.3f}")
if lines:
    open(os.path.join(out, tag + "_pmc_calibration.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
