#!/bin/bash
# round 5, GPU call D: the queue-priority A/B of the banded pipeline (+ hardware-queue count), host laps, full timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
L=$R/rust-bio_amd/libbiogpu.so
run() {  # tag, pairs, env...
    local tag=$1 n=$2; shift 2
    env "$@" timeout 200 python tools/exp/time_banded.py $L $n 2>&1 | grep -v amdgpu.ids | sed "s/^/$tag  /" | tee -a $O/summary.txt
}
for rep in 1 2; do
    run "prio=1      " 65536 BG_BAND_STREAM_PRIO=1
    run "prio=0      " 65536 BG_BAND_STREAM_PRIO=0
done
run "prio=0 hwq=8" 65536 BG_BAND_STREAM_PRIO=0 GPU_MAX_HW_QUEUES=8
run "prio=1 hwq=8" 65536 BG_BAND_STREAM_PRIO=1 GPU_MAX_HW_QUEUES=8
run "prio=1 100k " 100000 BG_BAND_STREAM_PRIO=1
run "prio=0 100k " 100000 BG_BAND_STREAM_PRIO=0
BG_TRACE=1 BG_BAND_STREAM_PRIO=1 timeout 200 python tools/exp/time_banded.py $L 65536 > $O/laps_prio1.log 2>&1
BG_TRACE=1 BG_BAND_STREAM_PRIO=0 timeout 200 python tools/exp/time_banded.py $L 65536 > $O/laps_prio0.log 2>&1
BG_BAND_STREAM_PRIO=1 timeout 300 bash tools/exp/trace_banded_full.sh 65536 > $O/timeline_prio1.txt 2>&1
BG_BAND_STREAM_PRIO=0 timeout 300 bash tools/exp/trace_banded_full.sh 65536 > $O/timeline_prio0.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_banded.py -m gpu -x -q 2>&1 | tail -2 | tee -a $O/summary.txt
