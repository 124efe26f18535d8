# Bench lines over (steps, sets per launch, launches in flight); run on the GPU box from the
# repository root.  Usage: bash tools/sweep_sets.sh "20 200" "4 1;4 2;8 1"
set -e
KS=${1:-"20 200"}
SFS=${2:-"4 1;4 2;4 3;8 1;8 2;2 3"}
OUT=gpurun_out/sweep_sets.txt
: > $OUT
IFS=';' read -ra LIST <<< "$SFS"
for K in $KS; do
for sf in "${LIST[@]}"; do
  set -- $sf
  timeout -k 10 100 python bench.py --no-cpu-baseline --steps $K --warmup 5 --sets-per-launch $1 --in-flight $2 --repeats 15 ${EXTRA} > gpurun_out/sw.json
  python3 - "$K $sf" <<'PY' >> $OUT
import json, sys
d = json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1])
k = d["kernels_ms"]; r = d["roofline"]
print("K S F = %-10s %10.1f img/s step %.5f (min %.5f) front %.5f nms %.5f gather %.5f frac %.4f alone %.4f cached %.4f enq %.2f" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], k["front"], k["segment_nms"], k["gather"], r["frac"], r["frac_alone"], r["frac_cache_resident"], d["config"]["host_enqueue_us_per_step"]))
PY
done; done
cat $OUT
