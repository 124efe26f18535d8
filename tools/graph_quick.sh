# dense-path parity on the GPU, then the per-kernel times of the large-segment configurations
# (29 500 / 2 000 / clustered candidates per segment, 32 images of 1920x1280); stops at the first failure
set -e
OUT=gpurun_out/graph; mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_u_dense_nms.py -x -q -m gpu --timeout 200 --timeout-method thread > $OUT/parity.log 2>&1 || { tail -40 $OUT/parity.log; exit 1; }
tail -n 1 $OUT/parity.log
: > $OUT/breakdown.txt
CONF=0.001 DENSE=1 timeout -k 10 120 python3 tools/nms_breakdown.py >> $OUT/breakdown.txt 2>$OUT/err.txt
DENSE=1 CONF=0.5 timeout -k 10 120 python3 tools/nms_breakdown.py >> $OUT/breakdown.txt 2>>$OUT/err.txt
DENSE=1 CONF=0.5 CLUSTERED=1 timeout -k 10 120 python3 tools/nms_breakdown.py >> $OUT/breakdown.txt 2>>$OUT/err.txt
cat $OUT/breakdown.txt
