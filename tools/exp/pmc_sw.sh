#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in WRITE_SIZE FETCH_SIZE; do
rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmcsw_$C -o x -- python $R/bench.py --skip-cpu --skip-fm --skip-banded --skip-pipeline --steps 2 --warmup 0 > /tmp/pmcsw_$C.log 2>&1
python - $C <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pmcsw_%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if "bgsw" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(bgsw")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(sys.argv[1], k, "mean per launch: %.3f G (raw counter units)" % (sum(v.values()) / len(v) / 1e9), len(v))
PY
done
