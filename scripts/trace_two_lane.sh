#!/bin/bash
# kernel trace of scripts/two_lane_graph.py (do consecutive steps on two streams of one graph overlap?)  -> gpurun_out/two_lane_trace.txt
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/tl_kt" -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/two_lane_graph.py 20 2000 ${1:-2} 2 > "$OUT/tl_kt.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/tl_kt/kt_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(f"{out}/two_lane_trace.txt", "w") as f:
    for r in rows[-260:]:
        f.write("%10.2f %10.2f  q%s  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:50]))
PY
tail -3 "$OUT/tl_kt.log"
