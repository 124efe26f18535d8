# Parity tests of the detection path followed by the bench lines used to compare kernel
# variants (cfg2 with 3 and 1 batches in flight, cfg3, cfg4).  Run on the GPU box from the
# repository root; results land in gpurun_out/quick_eval.txt.
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread > gpurun_out/quick_parity.log 2>&1
tail -1 gpurun_out/quick_parity.log
: > gpurun_out/quick_eval.txt
for args in "--config cfg2" "--config cfg2" "--config cfg2 --in-flight 1 --sets-per-launch 1" "--config cfg3 --steps 60" "--config cfg4"; do
  timeout -k 10 120 python bench.py --no-cpu-baseline $args > gpurun_out/quick_one.json
  python3 - "$args" <<'PY' >> gpurun_out/quick_eval.txt
import json, sys
d = json.loads(open("gpurun_out/quick_one.json").read().strip().splitlines()[-1])
k = d["kernels_ms"]
print("%-28s %10.1f images/s  step %.5f ms  front %.5f nms %.5f gather %.5f  frac %.4f" % (
    sys.argv[1], d["value"], d["ms_per_step"], k["front"], k["segment_nms"], k["gather"], d["roofline"]["frac"]))
PY
done
cat gpurun_out/quick_eval.txt
