#!/bin/bash
# SQ counters of fmd_smems_kernel (tools/exp/time_fmd.py's workload): bash tools/exp/sq_fmd.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/sqf; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/p$i" -o sq -- python $R/tools/exp/time_fmd.py 20000000 200000 > "$OUT/p$i.log" 2>&1
done
grep -v amdgpu $OUT/p1.log | tail -2
python - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fmd_smems" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, per in sorted(acc.items()):
    v = list(per.values()); print("   %-24s %12.3f M per launch (%d)" % (c, sum(v) / len(v) / 1e6, len(v)))
PY
