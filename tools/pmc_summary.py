"""Summarise the rocprofv3 PMC passes written by collect_profiles.sh.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch (summed over the XCDs).  The
calibration kernels (tools/pmc_calib.hip) move exactly known byte counts in the access shapes of the engine
(random 64-byte blocks per quad, random 128-byte lines per quad / per 8 lanes, coalesced 4 / 8 / 16 B per lane
loads and stores); the ratio reported / moved of each shape is measured in the same collection run and the
engine kernels of that shape are corrected by 1 / ratio (KERNEL_SHAPES below; MI355X_MICROARCH.md "HBM": gfx950
tallies 128-byte read requests at 64 B).  Every entry keeps `raw_bytes` (what rocprofv3 said), the shape and the
ratio applied; `mean_bytes` is the corrected figure bench.py reports as roofline.traffic.  Kernels without a
listed shape are left uncorrected and say so.  Single-byte strided stores are the known outlier (6.3x).
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

# ---- calibration (first: the traffic figures below are corrected with it)
cal_lines = []
CAL = {}  # (calibration kernel, counter) -> reported / moved
expect = {}
for f in glob.glob(os.path.join(out, "cal_*.log")):
    for ln in open(f):
        if ln.startswith("EXPECT"):
            _, name, counter, nbytes = ln.split()
            expect[(name, counter)] = float(nbytes)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(out, "cal_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"]) * 1024.0
    for k, v in agg.items():
        base = k.split("(")[0].replace("void ", "").strip()
        for (name, counter), nbytes in expect.items():
            if counter == c and name == base:
                mean = sum(v.values()) / len(v)
                CAL[(name, c)] = mean / nbytes
                cal_lines.append(f"{name:24s} {c:10s} measured {mean/1e9:9.4f} GB  expected {nbytes/1e9:9.4f} GB  ratio {mean/nbytes:6.3f}")
if cal_lines:
    open(os.path.join(out, tag + "_pmc_calibration.txt"), "w").write("\n".join(sorted(cal_lines)) + "\n")
    print("\n".join(sorted(cal_lines)))

# access shape of the engine kernels that carry a roofline: (kernel substring, FETCH shape, WRITE shape); first match wins
KERNEL_SHAPES = [
    ("fm_search_fast2x_kernel", "k_gather128x4", "k_store8"),       # two queries per quad on the 2-step blocks (round 5: the default)
    ("fm_search_fast_kernel<STEP2>", "k_gather128x4", "k_store8"),  # the last template argument: 128-byte lines, two LF steps each
    ("fm_search_fast_kernel", "k_quadload", "k_store8"),                                  # 1-step blocks: 64-byte lines
    ("fm_backward_search_kernel", "k_quadload", "k_store8"),
    ("sa_sampled_get", "k_quadload", "k_store8"), ("fmd_smems", "k_quadload", "k_store4"),
    ("banded_fill2p_kernel", "k_load16", "k_store16"),  # uint4 chunk loads, uint4 traceback lines
    ("banded_fill2i_kernel", "k_load16", "k_store16"), ("banded_fill2_kernel", "k_load16", "k_store16"),
    ("banded_traceback_kernel", "k_quadload", "k_store8"), ("banded_epilogue_kernel", "k_load16", "k_store8"),
    ("band_rows_kernel", "k_load4", "k_store4"),
    ("sw_fill_pk16_kernel", "k_load4", "k_store8"), ("sw_fill_kernel", "k_load4", "k_store8"),
    ("sw_traceback_kernel", "k_quadload", "k_store8"),
    ("fq_fused_kernel", "k_load16", "k_store16"),  # the one-pass reader (round 6): uint4 tile loads, uint4 sequence / quality stores
    ("fq_gather_kernel", "k_load16", "k_store16"), ("fq_count_newlines", "k_load16", "k_store4"),
    ("fq_line_starts", "k_load16", "k_store8"), ("fq_line_info", "k_load8", "k_store8"), ("fq_measure", "k_load16", "k_store4"),
]


def corrected(kernel, counter, raw):
    """raw bytes of one counter -> (bytes, {"shape", "ratio"} or None)"""
    step2 = "fm_search_fast_kernel<" in kernel and kernel.strip().endswith(", true>")
    for sub, fs, ws in KERNEL_SHAPES:
        if (sub == "fm_search_fast_kernel<STEP2>" and step2) or sub in kernel:
            shape = fs if counter == "FETCH_SIZE" else ws
            ratio = CAL.get((shape, counter))
            if ratio and ratio > 0:
                return raw / ratio, {"shape": shape, "ratio": round(ratio, 4)}
            return raw, None
    return raw, None


def entry(kernel, counter, vals):
    raw = sum(vals) * 1024.0 / len(vals)
    cor, how = corrected(kernel, counter, raw)
    return {"launches": len(vals), "mean_bytes": cor, "raw_bytes": raw, "calibration": how}


# ---- HBM traffic
res = {"unit": "bytes per launch (mean over launches); mean_bytes = raw_bytes / the measured ratio of the kernel's access shape "
               "(tools/pmc_calib.hip, <tag>_pmc_calibration.txt); calibration null = no listed shape, uncorrected",
       "csrc_sha": SHA, "calibration": {f"{n} {c}": round(r, 4) for (n, c), r in sorted(CAL.items())}, "kernels": {}, "launch_shape": shape}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in counters.get(c, {}).items():
        res["kernels"].setdefault(k, {})[c] = entry(k, c, v)
# FASTQ ingest is many short kernels per call: bytes of all of them per bg_fastq_parse_dev call (the leg makes 4 calls)
fq = sum(corrected(k, c, sum(v) * 1024.0)[0] for c in ("FETCH_SIZE", "WRITE_SIZE") for k, v in counters.get(c, {}).items() if k.startswith("fq_"))
res["ingest_bytes_per_call"] = fq / 4.0
# the FM search kernel on the index beyond the Infinity Cache (its own passes)
big = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, vs in per_kernel(os.path.join(out, "big_" + c)).get(c, {}).items():
        if ("fm_search_fast_kernel" in k or "fm_search_fast2x_kernel" in k) and "_kernel<true" not in k and (c not in big or len(vs) > big[c]["launches"]):
            big[c] = entry(k, c, vs)   # the byte flavour of the whole-pattern search (the SEEDS flavour belongs to seed_extend)
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
SE_KERNELS = ("se_", "sa_sampled_get", "sa_raw", "interval_rows", "fm_search_fast_kernel<true", "fm_search_fast2x_kernel<true", "fm_backward_search_kernel<false, true",
              "semiglobal", "sw_fill_pk16", "sw_traceback_kernel")
se = {"kernels": {}}
tot_bytes = 0.0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, vs in per_kernel(os.path.join(out, "big_" + c)).get(c, {}).items():
        if not k.startswith(("at::", "void at::", "elementwise_kernel", "rocprim", "void rocprim")) and any(t in k for t in SE_KERNELS):
            e = entry(k, c, vs)
            e["total_bytes"] = e["mean_bytes"] * len(vs)
            se["kernels"].setdefault(k, {})[c] = e
            tot_bytes += e["total_bytes"]
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

