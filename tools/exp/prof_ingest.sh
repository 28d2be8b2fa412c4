#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ing -o x -- python $R/bench.py --skip-cpu --skip-fm --skip-banded --skip-pipeline --skip-k1 --skip-semiglobal --skip-packed --pairs 65536 --steps 1 --warmup 0 > /tmp/ing.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/ing/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f))):
        if "fq_" in r["Name"] or "cigar" in r["Name"]:
            print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
