#!/bin/bash
# Round artefacts for profiles/ (run on the GPU box from the repo root):  bash scripts/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench workload   -> gpurun_out/<tag>_kernel_stats.csv
#   2. rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (never together with
#      other trace domains)                                               -> gpurun_out/<tag>_pmc_{fetch,write}.csv
#   3. traffic per step summed over the kernels of the solve               -> gpurun_out/<tag>_traffic.json
set -e
TAG=${1:-r01_panel}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra --steps 10 --warmup 3 --in-flight 1"   # one step at a time: per-kernel durations without overlap
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_kt" -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > "$OUT/${TAG}_kt.log" 2>&1
cp "$OUT/${TAG}_kt/kt_kernel_stats.csv" "$OUT/${TAG}_kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  c=$(echo $C | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --kernel-trace --pmc $C -d "$OUT/${TAG}_pmc_$c" -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > "$OUT/${TAG}_pmc_$c.log" 2>&1
  cp "$OUT/${TAG}_pmc_$c/pmc_counter_collection.csv" "$OUT/${TAG}_pmc_$c.csv"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = {}
for c, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f"{out}/{tag}_pmc_{c}.csv")):
        if row["Counter_Name"] == name:
            per[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    res[name] = {k: {"launches": len(v), "KiB_per_launch": sum(v) / len(v)} for k, v in per.items()}
solve = ["fmpc_cold_panel", "fmpc_cold_dz", "void fmpc_newton_wave<27>"]
def tot(name, factor):
    return sum(res[name].get(k, {"KiB_per_launch": 0.0})["KiB_per_launch"] for k in solve) * 1024 * factor
doc = {"hbm_bytes_per_launch": tot("FETCH_SIZE", 2.0) + tot("WRITE_SIZE", 1.0),
       "note": "one 'launch' = the kernels of one solve of 2000 problems (fmpc_cold_panel + fmpc_cold_dz + the decision pass of "
               "fmpc_newton_wave<27>); rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB units; FETCH_SIZE doubled "
               "(scripts/fetch_calib.hip: it reads 1/2 of the bytes on gfx950), WRITE_SIZE exact",
       "detail": res}
json.dump(doc, open(f"{out}/{tag}_traffic.json", "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("hbm_bytes_per_launch",)}))
PY
cat "$OUT/${TAG}_kernel_stats.csv" | cut -c1-140
