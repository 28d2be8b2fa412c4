#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trb -o x -- python $R/tools/exp/time_banded.py $R/rust-bio_amd/libbiogpu.so ${1:-49152} > /tmp/trb.log 2>&1
tail -1 /tmp/trb.log
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/trb/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "bgband" in n or "banded" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.replace("(anonymous namespace)::", "").replace("void ", "").replace("bgband_dev::", "").split("(")[0][:28], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# last call = last third of the launches
k = len(rows) // 3
rows = rows[k:2 * k]
t0 = rows[0][0]
busy = 0
cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, n, q in rows:
    print("%9.2f %9.2f  %-28s q=%s" % ((s - t0) / 1e6, (e - t0) / 1e6, n, q))
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.1f ms, device busy (union) %.1f ms" % ((rows[-1][1] - t0) / 1e6, busy / 1e6))
PY
