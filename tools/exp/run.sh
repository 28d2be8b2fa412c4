#!/bin/bash
# One parameterised recipe for the GPU calls of a round (round 6 on; the one-off r05_run_*.sh scripts are history:
# docs/history/).  Usage, through gpurun, from the repo root:
#     gpurun --timeout N -- 'bash tools/exp/run.sh <tag> <step> [<step> ...]'
# Every step appends its last line(s) to gpurun_out/<tag>/summary.txt and keeps its full log next to it.  Steps:
#   probe                 host cores / memory / cgroup limits / GPU memory of the box
#   tests                 pytest -m gpu (the whole suite)        tests:<expr>  pytest -m gpu -k <expr>
#   smoke                 __graft_entry__.smoke()
#   fuzz_fm:<seed>:<s>    tests/fuzz_fm.py            fuzz_fm_wide:<seed>:<s>   the same on the forced 64-bit layout
#   fuzz_banded:<seed>:<s>[:k3p]   fuzz_banded_long:<seed>:<s>:<len>   fuzz_pairwise:<seed>:<s>   fuzz_pk16:<seed>:<s>   fuzz_fastq:<seed>:<s>   fuzz_pipeline:<seed>:<s>
#   bench[:args]          python bench.py [args ...] (args separated by ',')  -> <tag>/bench.json
#   profiles              tools/collect_profiles.sh <tag>  (kernel stats + PMC passes of the default bench command)
#   py:<script>[:args]    python <script> [args ...] (args separated by ',')  -> <tag>/<script basename>.json (stdout)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:?tag}; shift
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
export PYTHONUNBUFFERED=1
say() { echo "$*" | tee -a "$O/summary.txt"; }
say "# $(date -u +%FT%TZ) csrc $(python tools/csrc_hash.py) steps: $*"
last() { grep -v amdgpu.ids "$1" | tail -${2:-1}; }
for step in "$@"; do
  IFS=: read -r what a b c <<< "$step"
  case $what in
    probe)
      { echo "nproc $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null; free -g | head -2;
        python - <<'PY'
import torch
f, t = torch.cuda.mem_get_info()
print("hbm free/total GB", round(f / 1e9, 1), round(t / 1e9, 1), torch.cuda.get_device_name(0))
PY
      } > "$O/probe.log" 2>&1; say "probe: $(grep -v amdgpu.ids "$O/probe.log" | tr '\n' ' ')" ;;
    tests)
      if [ -n "$a" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$a" > "$O/tests_$a.log" 2>&1; say "tests -k $a: $(last "$O/tests_$a.log")";
      else timeout 1500 python -m pytest tests -m gpu -q > "$O/tests.log" 2>&1; say "tests: $(last "$O/tests.log")"; grep -E "^(FAILED|ERROR)" "$O/tests.log" | head -20 | tee -a "$O/summary.txt"; fi ;;
    smoke) timeout 600 python __graft_entry__.py smoke > "$O/smoke.log" 2>&1; say "smoke: $(last "$O/smoke.log")" ;;
    fuzz_fm) timeout $((b + 200)) python tests/fuzz_fm.py "$a" "$b" > "$O/fuzz_fm_$a.log" 2>&1; say "fuzz_fm $a $b: $(last "$O/fuzz_fm_$a.log")" ;;
    fuzz_fm_wide) timeout $((b + 200)) python tests/fuzz_fm.py "$a" "$b" wide > "$O/fuzz_fm_wide_$a.log" 2>&1; say "fuzz_fm wide $a $b: $(grep -E "MISMATCH|Error|error" "$O/fuzz_fm_wide_$a.log" | head -3) $(last "$O/fuzz_fm_wide_$a.log")" ;;
    fuzz_banded) timeout $((b + 200)) python tests/fuzz_banded.py "$a" "$b" $c > "$O/fuzz_banded_$a.log" 2>&1; say "fuzz_banded $a $b $c: $(last "$O/fuzz_banded_$a.log")" ;;
    fuzz_banded_long) timeout $((b + 300)) python tests/fuzz_banded_long.py "$a" "$b" "$c" > "$O/fuzz_banded_long_$a.log" 2>&1; say "fuzz_banded_long $a $b $c: $(last "$O/fuzz_banded_long_$a.log")" ;;
    fuzz_pairwise) timeout $((b + 200)) python tests/fuzz_pairwise.py "$a" "$b" > "$O/fuzz_pairwise_$a.log" 2>&1; say "fuzz_pairwise $a $b: $(last "$O/fuzz_pairwise_$a.log")" ;;
    fuzz_fastq) timeout $((b + 200)) python tests/fuzz_fastq.py "$a" "$b" > "$O/fuzz_fastq_$a.log" 2>&1; say "fuzz_fastq $a $b: $(grep MISMATCH "$O/fuzz_fastq_$a.log" | head -3) $(last "$O/fuzz_fastq_$a.log")" ;;
    fuzz_pipeline) timeout $((b + 200)) python tests/fuzz_pipeline.py "$a" "$b" > "$O/fuzz_pipeline_$a.log" 2>&1; say "fuzz_pipeline $a $b: $(grep -E "MISMATCH|Error" "$O/fuzz_pipeline_$a.log" | head -3) $(last "$O/fuzz_pipeline_$a.log")" ;;
    fuzz_pk16) timeout $((b + 200)) python tests/fuzz_pk16.py "$a" "$b" > "$O/fuzz_pk16_$a.log" 2>&1; say "fuzz_pk16 $a $b: $(last "$O/fuzz_pk16_$a.log")" ;;
    bench) timeout 1500 python bench.py ${a//,/ } > "$O/bench.json" 2> "$O/bench.err"; say "bench ${a}: $(head -c 600 "$O/bench.json")" ;;
    profiles) timeout 3000 bash tools/collect_profiles.sh "$TAG" > "$O/profiles.log" 2>&1; say "profiles: $(ls gpurun_out/profiles_$TAG 2>/dev/null | tr '\n' ' ')" ;;
    py) n=$(basename "$a" .py); timeout ${PY_TIMEOUT:-1500} python "$a" ${b//,/ } > "$O/$n.json" 2> "$O/$n.err"; say "py $a $b: rc $? $(head -c 1500 "$O/$n.json")"; grep -v amdgpu.ids "$O/$n.err" | tail -5 | tee -a "$O/summary.txt" ;;
    *) say "unknown step $step" ;;
  esac
done
