set -e
YP_IMG_VARIANT=4 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 200 --timeout-method thread -k "v3 and image" 2>&1 | tail -1
for v in 0 2 4; do
for f in 2 1; do
YP_IMG_VARIANT=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --config cfg4 --repeats 6 --in-flight $f > gpurun_out/sweep.json
python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels_ms']; r=d['roofline']; print('  cfg4 variant %s in-flight %s: %.0f images/s, step %.2f us; per launch front %.1f nms %.1f gather %.1f us (alone %.1f %.1f %.1f); frac %.3f' % (sys.argv[2], sys.argv[3], d['value'], 1e3*d['ms_per_step'], 1e3*k['front'], 1e3*k['segment_nms'], 1e3*k['gather'], 1e3*k['alone']['front'], 1e3*k['alone']['segment_nms'], 1e3*k['alone']['gather'], r['frac']))" gpurun_out/sweep.json $v $f
done
done
