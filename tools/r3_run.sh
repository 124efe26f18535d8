# one GPU session of round 3: GPU tests, default bench, per-configuration lines, SQ counters
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/${1:-r3}
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["kernels_ms"], d["roofline"]["frac"], d.get("nms_roofline",{}).get("frac"))
PY
for cfg in cfg1 cfg4 "cfg2 --clustered" cfg3; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --config $cfg --steps 40 --warmup 8 --repeats 5 >> $OUT/configs.jsonl 2>> $OUT/configs.err
done
python3 - $OUT/configs.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d["config"]["workload"][:40], round(d["value"]), d["ms_per_step"], d["kernels_ms"].get("front"), d["kernels_ms"].get("segment_nms"), d["kernels_ms"].get("gather"))
PY
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 8 --repeats 2 > $OUT/pmc_sq.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/pmc_sq/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'][:50]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k in acc:
    if 'k_' in k: print(k, n[k], {c: round(v/max(n[k],1)) for c,v in acc[k].items()})
PY
