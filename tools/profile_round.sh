# Collects the profiles that back the bench line: rocprofv3 kernel statistics of the default
# bench command and, in separate passes, the HBM traffic counters of the same command; then the
# large-segment NMS path at cfg5 and cfg3 (kernel statistics + per-kernel HIP-event breakdown) and
# one bench line over RCCL with a world of one rank.  Run on the GPU box from the repository root;
# results land under gpurun_out/prof_final (copy what is to be kept into profiles/).
# Every GPU step runs under its own time limit and the script stops at the first failure.
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
# large segments: graph path
for cfg in cfg5 cfg3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -- python3 bench.py --no-cpu-baseline --config $cfg --steps 8 --warmup 2 --repeats 3 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  find $OUT/stats_$cfg -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$cfg.csv \;
done
CONF=0.001 timeout -k 10 120 python3 tools/nms_breakdown.py > $OUT/nms_breakdown.txt 2>&1
CONF=0.5 timeout -k 10 120 python3 tools/nms_breakdown.py >> $OUT/nms_breakdown.txt 2>&1
CONF=0.5 CLUSTERED=1 timeout -k 10 120 python3 tools/nms_breakdown.py >> $OUT/nms_breakdown.txt 2>&1
cat $OUT/nms_breakdown.txt
# the multi-rank code path (process group, barriers, one all-gather per launch) with one rank on RCCL
YP_BENCH_FORCE_DIST=1 YP_BENCH_WATCHDOG=120 timeout -k 10 200 python3 bench.py --no-cpu-baseline > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
tail -1 $OUT/bench_rccl_world1.json | cut -c1-300
