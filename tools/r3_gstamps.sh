set -e
export YOLOPOST_LIB=$PWD/2d_object_detection_amd/libyolopost_stamps.so
OUT=gpurun_out/gstamps.txt; : > $OUT
CONF=0.5 timeout -k 10 120 python3 tools/graph_stamps.py >> $OUT 2>gpurun_out/gstamps.err
CONF=0.001 timeout -k 10 120 python3 tools/graph_stamps.py >> $OUT 2>>gpurun_out/gstamps.err
CONF=0.5 CLUSTERED=1 timeout -k 10 120 python3 tools/graph_stamps.py >> $OUT 2>>gpurun_out/gstamps.err
cat $OUT
