set -e
timeout -k 10 600 python -u -m pytest tests/test_u_dense_nms.py -x -q -m gpu --timeout 200 --timeout-method thread 2>&1 | tail -2
CONF=0.001 timeout -k 10 120 python3 tools/nms_breakdown.py
DENSE=1 CONF=0.5 timeout -k 10 120 python3 tools/nms_breakdown.py
DENSE=1 CONF=0.5 CLUSTERED=1 timeout -k 10 120 python3 tools/nms_breakdown.py
DENSE=1 B=1 CONF=0.001 timeout -k 10 120 python3 tools/nms_breakdown.py
