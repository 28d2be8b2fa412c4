#!/bin/bash
# round 5, GPU call A: tests, chain A/B + timeline, calibration ratios, offline banded parity, long-read fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
timeout 300 python tools/exp/time_banded_ab.py 65536 band_chain_rows=0:1 > $O/chain_ab.log 2>&1; cat $O/chain_ab.log | tee -a $O/summary.txt
timeout 300 bash tools/exp/trace_banded.sh 65536 > $O/timeline_chain_rows.txt 2>&1; tail -60 $O/timeline_chain_rows.txt | head -70
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/cal_$C -o cal -- $R/tools/pmc_calib > $O/cal_$C.log 2>&1
done
cd $R
python - <<'PY' | tee -a gpurun_out/r05a/summary.txt
import csv, glob, collections, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r05a")
exp = {}
for f in glob.glob(O + "/cal_*.log"):
    for ln in open(f):
        if ln.startswith("EXPECT"):
            _, n, c, b = ln.split(); exp[(n, c)] = float(b)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(O + "/cal_" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Dispatch_Id"]] += float(r["Counter_Value"]) * 1024
    for k, v in sorted(agg.items()):
        if (k, c) in exp:
            m = sum(v.values()) / len(v)
            print("CAL %-16s %-10s measured %8.4f GB expected %8.4f GB ratio %.3f" % (k, c, m / 1e9, exp[(k, c)] / 1e9, m / exp[(k, c)]))
PY
rm -rf $O/cal_FETCH_SIZE $O/cal_WRITE_SIZE
timeout 600 python tools/exp/banded_offline_parity.py 100000 4096 > $O/r05_banded_offline_parity.json 2> $O/offline.err; cat $O/r05_banded_offline_parity.json | tee -a $O/summary.txt; tail -3 $O/offline.err
timeout 300 python tests/fuzz_banded_long.py 20260925 150 12000 > $O/fuzz_long.log 2>&1; tail -8 $O/fuzz_long.log | tee -a $O/summary.txt
