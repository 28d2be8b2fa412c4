#!/bin/bash
# SQ issue / stall counters of ONE kernel of the bench (separate rocprofv3 --pmc passes, kernel-trace only):
#     bash tools/exp/sq_kernel.sh <kernel name substring> <bench.py args ...>    -> stdout, values per launch in millions
K=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/sqk
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/p$i" -o sq -- python $R/bench.py --skip-cpu --steps 2 --warmup 0 "$@" > "$OUT/p$i.log" 2>&1
done
python - "$OUT" "$K" <<'PY'
import csv, glob, sys, collections
out, key = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, per in sorted(cs.items()):
        v = list(per.values())
        print("   %-24s %12.3f M per launch (%d launches)" % (c, sum(v) / len(v) / 1e6, len(v)))
PY
