#!/bin/bash
# Round-2 artefacts for profiles/ (run on the GPU box from the repo root):  bash scripts/prof_round2.sh <tag> <target...>
# Per target (scripts/prof_target.py):
#   1. rocprofv3 --kernel-trace --stats                                   -> gpurun_out/<tag>_<target>_kernel_stats.csv
#   2. rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, SEPARATE passes -> ..._pmc_{fetch,write}.csv
#   3. SQ counter passes (8 counters each, --pmc only with --kernel-trace)     -> ..._sq.txt
#   4. HBM bytes per launch and kernel (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, KiB units) -> ..._traffic.json
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for TGT in "$@"; do
  P="$OUT/${TAG}_${TGT}"
  rocprofv3 --kernel-trace --stats -d "${P}_kt" -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_target.py $TGT 10 > "${P}_kt.log" 2>&1
  cp "${P}_kt/kt_kernel_stats.csv" "${P}_kernel_stats.csv"
  for C in FETCH_SIZE WRITE_SIZE; do
    c=$(echo $C | tr A-Z a-z | cut -d_ -f1)
    rocprofv3 --kernel-trace --pmc $C -d "${P}_pmc_$c" -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_target.py $TGT 4 > "${P}_pmc_$c.log" 2>&1
    cp "${P}_pmc_$c/pmc_counter_collection.csv" "${P}_pmc_$c.csv"
  done
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C -d "${P}_sq$i" -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_target.py $TGT 3 > "${P}_sq$i.log" 2>&1
  done
  python3 - "$OUT" "$TAG" "$TGT" <<'PY'
import csv, glob, json, sys, collections
out, tag, tgt = sys.argv[1:4]
P = f"{out}/{tag}_{tgt}"
def short(k):
    k = k.split("(")[0]
    return k[5:] if k.startswith("void ") else k
res = {}
for c, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f"{P}_pmc_{c}.csv")):
        if row["Counter_Name"] == name:
            per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    res[name] = {k: {"launches": len(v), "KiB_per_launch": sum(v) / len(v)} for k, v in per.items()}
kern = {}
most = max([v["launches"] for k, v in res["FETCH_SIZE"].items() if k.startswith("fmpc")] or [0])
for k in res["FETCH_SIZE"]:
    if not k.startswith("fmpc"):
        continue                                      # (prefix match: template arguments and torch fill kernels differ)
    if 2 * res["FETCH_SIZE"][k]["launches"] < most:
        continue                                      # one-off launches (factor export, the panel kernel building J): not part of a solve
    f = res["FETCH_SIZE"][k]["KiB_per_launch"] * 1024 * 2.0
    w = res["WRITE_SIZE"].get(k, {"KiB_per_launch": 0.0})["KiB_per_launch"] * 1024
    kern[k] = {"hbm_bytes_per_launch": f + w, "fetch_bytes": f, "write_bytes": w}
doc = {"tag": tag, "target": tgt, "kernels": kern, "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] for v in kern.values()),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB units; FETCH_SIZE doubled (MI355X_MICROARCH.md: it reads "
               "1/2 of the bytes on gfx950), WRITE_SIZE exact; hbm_bytes_per_launch = sum over the fmpc_* kernels of ONE solve"}
json.dump(doc, open(f"{P}_traffic.json", "w"), indent=1)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{P}_sq*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(f"{P}_sq.txt", "w") as fh:
    for k, d in agg.items():
        if not k.startswith("fmpc"):
            continue
        fh.write(k + "\n")
        for c, v in sorted(d.items()):
            fh.write("   %-34s n=%3d  mean %.6g\n" % (c, len(v), sum(v) / len(v)))
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in kern.items()}), "MB per launch")
print(open(f"{P}_sq.txt").read())
PY
  cut -c1-150 "${P}_kernel_stats.csv" | head -8
done
