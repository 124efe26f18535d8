# One bench line per configuration (kernel times per launch); run on the GPU box from the
# repository root.  Usage: bash tools/bench_configs.sh "cfg2 cfg3 cfg5" [extra bench flags]
set -e
CFGS=${1:-"cfg1 cfg2 cfg3 cfg3x2 cfg4 cfg5"}
shift || true
OUT=gpurun_out/bench_configs.txt
: > $OUT
: > gpurun_out/bench_configs.jsonl
for c in $CFGS; do
  extra=""
  case $c in *-clustered) extra="--clustered";; esac
  timeout -k 10 300 python bench.py --no-cpu-baseline --config ${c%-clustered} $extra --steps 40 --warmup 4 --repeats 5 "$@" > gpurun_out/bc.json
  cat gpurun_out/bc.json >> gpurun_out/bench_configs.jsonl
  python3 - "$c" <<'PY' >> $OUT
import json, sys
d = json.loads(open("gpurun_out/bc.json").read().strip().splitlines()[-1])
k = d["kernels_ms"]; r = d["roofline"]; n = d["nms_roofline"]
print("%-16s %10.1f img/s step %.5f ms | per launch of %d: front %.4f nms %.4f gather %.4f (alone %.4f %.4f %.4f) | frac %.3f alone %.3f | surv/img %.0f kept %.0f pairs %d ties %d" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["config"]["batches_per_launch"], k["front"], k["segment_nms"], k["gather"],
    k["alone"]["front"], k["alone"]["segment_nms"], k["alone"]["gather"], r["frac"], r["frac_alone"],
    d["config"]["survivors_per_image"], d["config"]["kept_per_image"], k["nms_iou_pairs"], d["config"]["iou_threshold_ties_per_batch"]))
PY
done
cat $OUT
