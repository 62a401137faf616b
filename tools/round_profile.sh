#!/bin/bash
# round-end measurement on a GPU box: bench lines, rocprofv3 kernel stats and HBM-traffic counters
# (separate --pmc passes, never mixed with trace domains other than --kernel-trace).  Two parts, one gpurun call each:
#   tools/round_profile.sh r4 A   the bench lines, kernel stats, counters, traffic.json of THIS build and the line that quotes it
#   tools/round_profile.sh r4 B   one tile of the 2 / 4 / 8-tile split, the multi-rank rehearsals on one GPU, the fp32-storage study lines
TAG=${1:-r5}
PART=${2:-A}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
if [ $PART = A ]; then
  timeout -k 10 700 python bench.py --steps 20 --warmup 5 > $O/bench_basin2048.json 2> $O/bench_basin2048.err; echo "bench rc=$?"; tail -c 300 $O/bench_basin2048.json; echo
  python bench.py --workload seamount256 --steps 40 --no-cpu-baseline > $O/bench_seamount256.json 2>/dev/null
  python bench.py --workload basin1024 --steps 10 --no-cpu-baseline > $O/bench_basin1024.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_line_under_rocprofv3_kernel_trace.json 2> $O/stats.err
  # the same under rocprofv3 with the layout left as it comes: the CSV then holds ONE layout, and its average is the events' average
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_one_layout -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tune-placement > $O/bench_line_under_rocprofv3_kernel_trace_one_layout.json 2> $O/stats1.err
  for set in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$set -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_$set.log 2>&1
  done
  python3 tools/pmc_summarise.py $O > $O/pmc_summary.csv 2>&1 || true
  # profiles/traffic.json of THIS build (bench.py quotes roofline.traffic only when the build ids agree), then the line that carries it
  python3 tools/make_traffic_json.py $O/pmc_summary.csv > $O/traffic_make.log 2>&1 && cp profiles/traffic.json $O/traffic.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_basin2048_with_traffic.json 2>/dev/null; tail -c 400 $O/bench_basin2048_with_traffic.json; echo
  find $O -name "*kernel_stats.csv" | head -3
else
  # one tile of the N-tile split (and the single tile through the same tool, for the bounds of profiles/tile_probe_bounds.json: tools/make_tile_bounds.py).
  # Interior tiles (4, 8 tiles) under the REAL transport -- an RCCL communicator of one rank whose tile is its own neighbour: grouped ncclSend / ncclRecv on
  # both streams, no Python per round (RCCL prints a banner on stdout: the JSON line is the one that starts with a brace); 2 tiles have no interior tile
  # and run under the stand-in mover, which the 8-tile line is also given beside the RCCL one
  timeout -k 10 200 python tools/tile_probe.py --tiles 1 --rank 0 --steps 10 > $O/tile_probe_1tiles.json 2> $O/tile_1.err
  timeout -k 10 200 python tools/tile_probe.py --tiles 2 --rank 1 --steps 10 > $O/tile_probe_2tiles.json 2> $O/tile_2.err
  for n in 4 8; do timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) --steps 10 --rccl-self 2> $O/tile_$n.err | grep "^{" > $O/tile_probe_${n}tiles.json; tail -c 200 $O/tile_probe_${n}tiles.json; echo; done
  timeout -k 10 200 python tools/tile_probe.py --tiles 8 --rank 4 --steps 10 > $O/tile_probe_8tiles_stand_in_mover.json 2> $O/tile_8s.err
  timeout -k 10 200 python tools/tile_probe.py --tiles 8 --rank 0 --steps 10 > $O/tile_probe_8tiles_first_tile.json 2> $O/tile_8_0.err
  POM_TILE_GRID=2x4 timeout -k 10 200 python tools/tile_probe.py --tiles 8 --rank 3 --steps 10 --rccl-self 2> $O/tile_2x4.err | grep "^{" > $O/tile_probe_8tiles_2x4.json
  timeout -k 10 200 python tools/tile_probe.py --tiles 8 --rank 4 --steps 10 --rccl-self 2>/dev/null | grep "^{" > $O/tile_probe_8tiles_same_box_as_2x4.json
  # a MODEL of link latency: every message round holds its stream for 30 us; nine rounds hidden against six (POMGPU_RIM_RESULTS_MAIN) and two (POMGPU_RIM_MAIN)
  timeout -k 10 200 python tools/tile_probe.py --tiles 8 --rank 4 --steps 10 --round-us 30 --ab POMGPU_RIM_RESULTS_MAIN=1 > $O/tile_probe_8tiles_round30us_results_rounds.json 2> $O/tile_8_m1.err
  timeout -k 10 200 python tools/tile_probe.py --tiles 8 --rank 4 --steps 10 --round-us 30 --ab POMGPU_RIM_MAIN=1 > $O/tile_probe_8tiles_round30us_rim_rounds.json 2> $O/tile_8_m2.err
  # the REAL transport on the one GPU: an RCCL communicator of one rank whose tile is its own neighbour -- grouped ncclSend / ncclRecv on both streams
  # (RCCL prints a banner on stdout: the JSON line is the one that starts with a brace)
  for sw in RIM_MAIN RIM_RESULTS_MAIN; do timeout -k 10 300 python tools/tile_probe.py --tiles 8 --rank 4 --steps 10 --rccl-self --ab POMGPU_$sw=1 2> $O/tile_8_rccl_$sw.err | grep "^{" > $O/tile_probe_8tiles_rccl_self_$sw.json; done
  POM_BENCH_REHEARSE=1 timeout -k 10 500 python3 bench.py --gpus 2 --workload basin1024 --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearsal_2ranks_one_gpu_basin1024.json 2> $O/rehearse2.err; echo "rehearse2 rc=$?"
  POM_BENCH_REHEARSE=1 timeout -k 10 600 python3 bench.py --gpus 4 --workload basin1024 --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearsal_4ranks_one_gpu_basin1024.json 2> $O/rehearse4.err; echo "rehearse4 rc=$?"
  POM_BENCH_REHEARSE=1 timeout -k 10 600 python3 bench.py --gpus 4 --workload basin2048 --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearsal_4ranks_one_gpu_basin2048.json 2> $O/rehearse4b.err; echo "rehearse4 basin2048 rc=$?"
  timeout -k 10 300 python bench.py --storage f32 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_basin2048_f32_storage_study.json 2> $O/f32.err; echo "f32 rc=$?"
  POM_BENCH_REHEARSE=1 timeout -k 10 500 python3 bench.py --gpus 2 --storage f32 --workload basin1024 --steps 3 --warmup 1 --no-cpu-baseline --no-alternate > $O/rehearsal_2ranks_one_gpu_basin1024_f32_storage.json 2> $O/rehearse2f32.err; echo "rehearse2 f32 rc=$?"
fi
ls $O
