#!/bin/bash
# SQ issue/stall counters of the full-matrix aligner's kernels (separate rocprofv3 --pmc passes, kernel-trace only):
#   tools/sq_counters.sh <tag>   -> gpurun_out/sq_<tag>.txt   (values per launch, in millions)
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --skip-cpu --skip-fm --skip-banded --skip-pipeline --steps 2 --warmup 0 ${SQ_BENCH_ARGS:-}"
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/p$i" -o sq -- $BENCH > "$OUT/p$i.log" 2>&1
done
python - "$OUT" > "$R/gpurun_out/sq_$TAG.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(bgsw")[0]
        if "bgsw" not in r["Kernel_Name"]:
            continue
        acc[k][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        per = collections.defaultdict(float)
        for d, x in v:
            per[d] += x
        print("   %-26s %12.1f M  (%d launches)" % (c, sum(per.values()) / len(per) / 1e6, len(per)))
PY
rm -rf "$OUT"
