#!/bin/bash
# SQ counter passes over the default bench workload (run on the GPU box from the repo root):
#   bash scripts/pmc_passes.sh <outdir-under-gpurun_out>
# One rocprofv3 run per pass (8 SQ slots); --pmc only ever together with --kernel-trace.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d "$OUT/p$i" -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 2 --in-flight 1 > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(d.items()):
            fh.write("   %-34s n=%3d  mean %.4g\n" % (c, len(v), sum(v) / len(v)))
print(open(out + "/summary.txt").read())
PY
