#!/bin/bash
# round 5, GPU call G: the join of c + 2 behind the chaining of c + 1, with K3p in 512- and 256-thread blocks
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
L=$R/rust-bio_amd/libbiogpu.so
run() { local tag=$1 n=$2; shift 2; timeout 200 python tools/exp/time_banded.py $L $n "$@" 2>&1 | grep -v amdgpu.ids | sed "s/^/$tag  /" | tee -a $O/summary.txt; }
for rep in 1 2; do
run "100k late512 " 100000
run "100k late256 " 100000 band_p_block256=1
run "100k early512" 100000 band_join_early=1
run "100k early256" 100000 band_join_early=1 band_p_block256=1
done
run "64k  late512 " 65536
run "64k  late256 " 65536 band_p_block256=1
timeout 300 bash tools/exp/trace_banded_full.sh 65536 > $O/timeline_late512.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_banded.py -m gpu -x -q 2>&1 | tail -2 | tee -a $O/summary.txt
