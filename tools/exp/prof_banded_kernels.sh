#!/bin/bash
# per-kernel times of one banded call of 100 000 x 10 kb pairs (rocprofv3 --kernel-trace --stats): bash tools/exp/prof_banded_kernels.sh <lib.so> [pairs]
R=${GRAFT_REPO_ROOT:-/root/repo}
SO=$(realpath $1); P=${2:-100000}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -o x -- python $R/tools/exp/time_banded.py $SO $P > /tmp/bk.log 2>&1
grep -v amdgpu /tmp/bk.log | tail -1
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/bk/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "band" in r["Name"] or "chain" in r["Name"] or "kmer" in r["Name"]]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
        print("   %-60s calls %4s avg %9.1f us total %8.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
