# SQ counters of the graph-path kernels at cfg3 and cfg5 (one GPU call; each pass its own run)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_graph; mkdir -p $OUT
for cfg in "0.5 cfg3" "0.001 cfg5"; do
  set -- $cfg
  CONF=$1 DENSE=1 N=4 timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/$2 -- python3 tools/nms_breakdown.py > $OUT/$2.log 2>&1 || { tail -20 $OUT/$2.log; exit 1; }
  CONF=$1 DENSE=1 N=4 timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $OUT/$2b -- python3 tools/nms_breakdown.py > $OUT/$2b.log 2>&1 || { tail -20 $OUT/$2b.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv,glob,collections,sys
for sub in ("cfg3","cfg3b","cfg5","cfg5b"):
    f=glob.glob(sys.argv[1]+'/'+sub+'/**/*counter_collection.csv',recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for fn in f:
        for r in csv.DictReader(open(fn)):
            k=r['Kernel_Name']
            if 'k_' not in k or 'at::' in k: continue
            k=k.split('(')[0][-40:]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
            if r['Counter_Name']=='SQ_WAVE_CYCLES': n[k]+=1
    print("==",sub)
    for k,v in acc.items():
        print(k, n[k], {c:round(x/max(1,n[k])) for c,x in v.items()})
PY
