set -e
bash tools/ab_image.sh run
YOLOPOST_LIB=$PWD/2d_object_detection_amd/libyolopost_stamps.so timeout -k 10 120 python3 tools/img_stamps.py
for v in base walk; do
YOLOPOST_LIB=$PWD/2d_object_detection_amd/libyolopost_$v.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --config cfg2 --clustered --steps 40 --warmup 8 --repeats 5 > gpurun_out/abi_$v.cfg2c.json
python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels_ms']; print('  %s cfg2-clustered: %.0f images/s, step %.2f us; per launch front %.1f nms %.1f gather %.1f us' % (sys.argv[2], d['value'], 1e3*d['ms_per_step'], 1e3*k['front'], 1e3*k['segment_nms'], 1e3*k['gather']))" gpurun_out/abi_$v.cfg2c.json $v
done
