for nf in 2 3 5 6 9 12 16; do
  echo "== in-flight $nf"
  timeout -k 10 120 python -u bench.py --no-cpu-baseline --in-flight $nf --steps 400 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['config']['host_enqueue_us_per_step'])"
done
