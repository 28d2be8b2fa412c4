#!/bin/bash
# per-kernel times of the FASTQ ingest leg (rocprofv3 --kernel-trace --stats): bash tools/exp/prof_ingest.sh [ctx option=value]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ing
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ing -o x -- python $R/bench.py --skip-cpu --skip-fm --skip-banded --skip-pipeline --skip-k1 --skip-semiglobal --skip-packed --fm-big-genome 0 --pairs 65536 --steps 3 --warmup 1 > /tmp/ing.log 2>&1
python - <<'PY'
import csv, glob, json
for f in glob.glob("/tmp/ing/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f))):
        if "fq_" in r["Name"] or "cigar" in r["Name"]:
            print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
try:
    d = json.loads([l for l in open("/tmp/ing.log") if l.startswith("{")][-1])
    print("ingest GB/s", d["ingest"]["value"])
except Exception as e:
    print("no bench line", e)
PY
