set -e
VARIANTS="base:" bash tools/ab_image.sh run
for v in base min9999; do echo == $v; YOLOPOST_LIB=$PWD/2d_object_detection_amd/libyolopost_$v.so timeout -k 10 200 python3 tools/latency_sweep.py; done
