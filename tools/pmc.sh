#!/bin/bash
# PMC counters per kernel for any command (run through gpurun): tools/pmc.sh <tag> <kernel-name filter> <python script> [args]
# Two counter passes (MFMA / waits, LDS / VALU), per-launch averages printed as a table.
set -o pipefail
TAG=$1; FLT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $SCRIPT "$@" > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
FLT="$FLT" OUT="$OUT" python3 - <<'PY'
import csv, glob, os, collections
out, flt = os.environ['OUT'], os.environ['FLT']
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, v in sorted(acc.items()):
        if flt in k:
            print(k, {c: float('%.4g' % (x / n[(k, c)])) for c, x in v.items()})
PY
