#!/bin/bash
# kernel + memory-copy timeline of the second warm banded call (every kernel, also the runtime's fill / copy kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trf
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trf -o x -- python $R/tools/exp/time_banded.py $R/rust-bio_amd/libbiogpu.so ${1:-49152} > /tmp/trf.log 2>&1
grep -v amdgpu.ids /tmp/trf.log | tail -3
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/trf/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.replace("(anonymous namespace)::", "").replace("void ", "").replace("bgband_dev::", "").split("(")[0][:34], "s" + str(r.get("Stream_Id", "?")) + "/q" + str(r.get("Queue_Id", "?"))))
for f in glob.glob("/tmp/trf/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?")[:22], "s" + str(r.get("Stream_Id", "?"))))
rows.sort()
band = [i for i, r in enumerate(rows) if "banded_traceback" in r[2]]
# three calls of the same shape, told apart by their last kernel (K4 of the last sub-batch)
k = len(band) // 3
lo = rows[band[k - 1]][1]      # end of call 1
hi = rows[band[2 * k - 1]][1]  # end of call 2
sel = [r for r in rows if r[0] > lo and r[1] <= hi + 1]
t0 = sel[0][0]
for s, e, n, q in sel:
    print("%9.2f %9.2f  %-34s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, n, q))
print("span %.1f ms" % ((sel[-1][1] - t0) / 1e6))
PY
