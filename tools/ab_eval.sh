# A/B comparison of two builds of the library on one box: alternates the bench between
# 2d_object_detection_amd/libyolopost_old.so and libyolopost_new.so (YOLOPOST_LIB selects it).
mkdir -p gpurun_out
: > gpurun_out/ab_eval.txt
for rep in 1 2 3; do
  for v in ${AB_VARIANTS:-old new}; do
    for args in "--config cfg2" "--config cfg4" ${AB_EXTRA:+"$AB_EXTRA"}; do
      YOLOPOST_LIB=$PWD/2d_object_detection_amd/libyolopost_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-large-input $args > gpurun_out/ab_one.json || exit 1
      python3 - "$v $args" <<'PY' >> gpurun_out/ab_eval.txt
import json, sys
d = json.loads(open("gpurun_out/ab_one.json").read().strip().splitlines()[-1])
k = d["kernels_ms"]
print("%-24s %10.1f images/s  in flight %d  front %.5f nms %.5f gather %.5f  alone %s" % (
    sys.argv[1], d["value"], d["config"]["batches_in_flight"], k["front"], k["segment_nms"], k["gather"], k.get("alone")))
PY
    done
  done
done
cat gpurun_out/ab_eval.txt
