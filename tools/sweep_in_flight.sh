# Throughput at cfg2 against the number of batches in flight, for several settings of HIP's
# GPU_MAX_HW_QUEUES (hardware queues per process).  Run on the GPU box from the repository root.
for q in ${QUEUES:-4 8}; do
  for nf in ${IN_FLIGHT:-3 4 5 6 7 8}; do
    YP_HW_QUEUES=$q timeout -k 10 100 python bench.py --no-cpu-baseline --no-large-input --in-flight $nf --steps 400 > gpurun_out/sw.json || exit 1
    python3 -c "
import json; d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); print('queues', $q, 'in flight', $nf, d['value'], d['ms_per_step'], d['config']['host_enqueue_us_per_step'])"
  done
done
