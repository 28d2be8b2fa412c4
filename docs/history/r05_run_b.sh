#!/bin/bash
# round 5, GPU call B: tests (+ the 64-bit FM tests), banded host-sync A/B + timeline, the 4.4 G-symbol FM run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -15 | tee -a $O/summary.txt
timeout 300 python tools/exp/time_banded_ab.py 65536 band_host_sync band_chain_rows=0:1 > $O/sync_ab.log 2>&1; grep -v amdgpu.ids $O/sync_ab.log | tee -a $O/summary.txt
timeout 300 bash tools/exp/trace_banded.sh 65536 > $O/timeline.txt 2>&1; grep -E "^ +[0-9]|^span" $O/timeline.txt | tail -45
timeout 900 python tools/exp/fm_wide_big.py 4400000000 10000000 > $O/r05_fm_wide_4g4.json 2> $O/fm_wide.err; echo "fm_wide rc $?" | tee -a $O/summary.txt
cat $O/r05_fm_wide_4g4.json | tee -a $O/summary.txt; tail -5 $O/fm_wide.err
