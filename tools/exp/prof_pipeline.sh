#!/bin/bash
# per-kernel times of the seed-and-extend leg alone (rocprofv3 --kernel-trace --stats): bash tools/exp/prof_pipeline.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o x -- python $R/bench.py --skip-cpu --skip-banded --skip-k1 --skip-semiglobal --skip-packed --skip-ingest --skip-fmd --fm-big-genome 0 --pairs 65536 --queries 100000 --steps 2 --warmup 1 > /tmp/pl.log 2>&1
python - <<'PY'
import csv, glob, json
for f in glob.glob("/tmp/pl/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.1f ms" % (tot / 1e6))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
        print("   %-70s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
try:
    d = json.loads([l for l in open("/tmp/pl.log") if l.startswith("{")][-1])
    print(json.dumps(d.get("seed_extend"))[:1500])
except Exception as e:
    print("no bench line", e); print(open("/tmp/pl.log").read()[-1500:])
PY
