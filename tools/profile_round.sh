# Collects the profiles that back the bench line: rocprofv3 kernel statistics of the default
# bench command and, in separate passes, the HBM traffic counters of the same command.
# Run on the GPU box from the repository root; results land under gpurun_out/.
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_final
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 > $OUT/pmc_$c.log 2>&1
done
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_summary.json
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
tail -1 $OUT/bench.json | cut -c1-300
head -8 $OUT/kernel_stats.csv
