set -e
for lib in libyolopost.so libyolopost_u4w6.so libyolopost_u4w8.so; do
  for nf in 1 3 4 6; do
    echo "== $lib in-flight $nf"
    YOLOPOST_LIB=$PWD/2d_object_detection_amd/$lib timeout -k 10 120 python -u bench.py --no-cpu-baseline --in-flight $nf | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['kernels_ms']['front'], d['kernels_ms']['segment_nms'], d['kernels_ms']['gather'])"
  done
done
