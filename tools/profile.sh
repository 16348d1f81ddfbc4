#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + PMC passes for bench.py, outputs under gpurun_out/<tag>/.
# usage: tools/profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-prof}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary $@"
echo "== kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || exit 1
grep '"metric"' $OUT/trace.log | tail -1
PARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-secondary --global-batch 16384 --chunk 16384 $@"
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1))
  echo "== pmc pass $i: $SET"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py $PARGS > $OUT/pmc$i.log 2>&1 || { tail -5 $OUT/pmc$i.log; exit 1; }
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.md && cat $OUT/summary.md
