# kernel statistics and SQ counters of the default bench (one GPU session)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/${1:-r3prof}
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline --repeats 5 > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -6 $OUT/kernel_stats.csv | cut -c1-200
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 8 --repeats 2 > $OUT/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 8 --repeats 2 > $OUT/pmc_sq2.log 2>&1 || true
python3 - $OUT <<'PY'
import csv,glob,collections,sys
for sub in ("pmc_sq","pmc_sq2"):
    f=glob.glob(sys.argv[1]+'/'+sub+'/**/*counter_collection.csv',recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
    for fn in f:
        for r in csv.DictReader(open(fn)):
            k=r['Kernel_Name'][:48]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
    for k in acc:
        if 'k_' in k: print(k, {c: round(v/max(n[k][c],1)) for c,v in acc[k].items()})
PY
