#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, no tracing beside --pmc) for one command.
#   scripts/pmc_collect.sh <out_dir> <kernel substring> -- python3 scripts/gemm_shapes.py one f16x3 38400 2048 512
# Writes <out_dir>/<group>/..._counter_collection.csv and prints the per-kernel mean of every counter.
set -e
out=$1; needle=$2; shift 3
export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
 "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum"
 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA_WRREQ_sum"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  # PMC_GROUPS="1 3": only these groups
  if [ -n "$PMC_GROUPS" ] && ! [[ " $PMC_GROUPS " == *" $i "* ]]; then i=$((i+1)); continue; fi
  d=$out/g$i; mkdir -p $d
  rocprofv3 --pmc $g -d $d --output-format csv -- "$@" > $d/run.log 2>&1 || { echo "group $i failed"; tail -3 $d/run.log; }
  i=$((i+1))
done
python3 - "$out" "$needle" <<'PY'
import csv, glob, sys, collections
out, needle = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if needle in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:62s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
