set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3a/pytest.log 2>&1
tail -3 gpurun_out/r3a/pytest.log
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/r3a/pmc_sq -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 8 --repeats 2 > gpurun_out/r3a/pmc_sq.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r3a/pmc_sq/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'][:60]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k in acc:
    print(k, n[k], {c: round(v/max(n[k],1)) for c,v in acc[k].items()})
PY
