#!/bin/bash
# round 5: a second round of differential fuzz (other seeds) on the final sources
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05fuzz2
mkdir -p $O
cd $R
echo "# csrc $(python tools/csrc_hash.py)" | tee $O/summary.txt
run() { local name=$1; shift; timeout 400 python "$@" > $O/$name.log 2>&1; echo "$* : $(grep -v amdgpu.ids $O/$name.log | tail -1)" | tee -a $O/summary.txt; }
run fm tests/fuzz_fm.py 20260929 170
run banded_k3p tests/fuzz_banded.py 20260929 170 k3p
run banded_long tests/fuzz_banded_long.py 20260929 170 12000
