#!/bin/bash
# per-kernel counters of the banded pipeline (kernel-trace + pmc only): tools/exp/pmc_banded.sh "<counters>" <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcb_$2
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d /tmp/pmcb_$2 -o x -- python $R/tools/exp/time_banded.py $R/rust-bio_amd/libbiogpu.so 16384 > /tmp/pmcb_$2.log 2>&1
python - $2 <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmcb_%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "bgband" in n or "banded" in n:
            n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("bgband_dev::", "").split("(")[0][:30]
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in sorted(agg.items()):
    print("%-32s" % n, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(cs.items())}, "(millions per launch)")
PY
