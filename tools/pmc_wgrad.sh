set -o pipefail
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/wg_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/wgrad_bench.py 8192 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/wg_pmc'
for f in sorted(glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','')[:40]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    for k,v in acc.items():
        if 'k_wgrad<' in k: print(k, {c: round(x/n[(k,c)]) for c,x in v.items()})
PY
