# A/B of compile-time variants of the per-image NMS kernel on the GPU box.
#     bash tools/ab_image.sh build      (here, on the CPU, before the gpurun call)
#     bash tools/ab_image.sh run        (on the GPU box, one call)
# Every variant first passes the parity tests of the YOLOv3 path, then is timed at cfg2 and cfg4.
set -e
VARIANTS=${VARIANTS:-"base:"}
CSRC=2d_object_detection_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function -Iinclude -I$CSRC"
if [ "$1" = "build" ]; then
  for v in $VARIANTS; do
    name=${v%%:*}; defs=$(echo "${v#*:}" | tr ',' ' ')
    rm -f 2d_object_detection_amd/libyolopost_$name.so
    python3 tools/kernel_resources.py --check tests/golden/kernel_resources.json $defs ||
        { echo "  $name: leaves the resource table, NOT built"; continue; }
    /opt/rocm/bin/hipcc $FLAGS $defs -shared -o 2d_object_detection_amd/libyolopost_$name.so \
        $CSRC/yolopost.hip $CSRC/gridcodec.hip $CSRC/decode_grad.hip $CSRC/metrics.hip 2>&1 | grep -v hip-link || true
    /opt/rocm/bin/hipcc $FLAGS $defs -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /dev/null $CSRC/yolopost.hip 2>&1 |
        grep -A12 "k_image_nms" | grep -E "VGPRs:|ScratchSize|Occupancy|LDS" | sed "s/.*remark: [^ ]* */  $name: /; s/ \[-Rpass.*//"
  done
  exit 0
fi
OUT=gpurun_out/ab_image.txt
: > $OUT
for v in $VARIANTS; do
  name=${v%%:*}
  export YOLOPOST_LIB=$PWD/2d_object_detection_amd/libyolopost_$name.so
  [ -f $YOLOPOST_LIB ] || { echo "== $name: not built (refused by the resource check)" | tee -a $OUT; continue; }
  echo "== $name" | tee -a $OUT
  timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 200 --timeout-method thread -k "v3" \
      > gpurun_out/abi_$name.parity.log 2>&1 || { head -n 40 gpurun_out/abi_$name.parity.log; echo "$name: parity FAILED, stopping" | tee -a $OUT; exit 1; }
  tail -n 1 gpurun_out/abi_$name.parity.log | tee -a $OUT
  for cfg in cfg2 cfg4 cfg1; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --config $cfg --repeats 8 > gpurun_out/abi_$name.$cfg.json 2>>gpurun_out/abi_$name.err || { head -n 40 gpurun_out/abi_$name.err; exit 1; }
    python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels_ms']; print('  %s: %.0f images/s, step %.2f us; per launch front %.1f nms %.1f gather %.1f us' % (sys.argv[2], d['value'], 1e3*d['ms_per_step'], 1e3*k['front'], 1e3*k['segment_nms'], 1e3*k['gather']))" gpurun_out/abi_$name.$cfg.json $cfg | tee -a $OUT
  done
done
