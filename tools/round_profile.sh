#!/bin/bash
# round-end measurement on a GPU box: bench lines, rocprofv3 kernel stats and HBM-traffic counters
# (separate --pmc passes, never mixed with trace domains other than --kernel-trace)
TAG=${1:-r2}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
python bench.py --side-config > $O/bench_basin2048_with_config1.json 2> $O/bench_basin2048.err; tail -c 300 $O/bench_basin2048_with_config1.json; echo
python bench.py --workload seamount256 --steps 20 > $O/bench_seamount256.json 2>/dev/null
python bench.py --workload basin1024 --steps 5 > $O/bench_basin1024.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$set -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_$set.log 2>&1
done
python3 tools/pmc_summarise.py $O > $O/pmc_summary.csv 2>&1 || true
# profiles/traffic.json of THIS build (bench.py quotes roofline.traffic only when the build ids agree), then the line that carries it
python3 tools/make_traffic_json.py $O/pmc_summary.csv > $O/traffic_make.log 2>&1 && cp profiles/traffic.json $O/traffic.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_basin2048_with_traffic.json 2>/dev/null; tail -c 400 $O/bench_basin2048_with_traffic.json; echo
POM_BENCH_REHEARSE=1 timeout -k 10 400 python3 bench.py --gpus 2 --workload basin1024 --steps 3 --warmup 1 > $O/rehearse2.json 2> $O/rehearse2.err; echo "rehearse2 rc=$?"
POM_BENCH_REHEARSE=1 timeout -k 10 400 python3 bench.py --gpus 4 --workload basin1024 --steps 3 --warmup 1 > $O/rehearse4.json 2> $O/rehearse4.err; echo "rehearse4 rc=$?"
find $O -name "*kernel_stats.csv" | head -3
ls $O
