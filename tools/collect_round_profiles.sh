#!/bin/bash
# after tools/round_profile.sh <tag> A and B (gpurun merges their output into gpurun_out/<tag>): copy what is judged into profiles/ under
# round-numbered names, profiles/traffic.json of the profiled build, and profiles/tile_probe_bounds.json (tools/make_tile_bounds.py)
TAG=${1:-r5}; N=${TAG#r}; O=gpurun_out/$TAG; P=profiles
set -e
for f in bench_basin2048 bench_basin2048_with_traffic bench_seamount256 bench_basin1024 bench_line_under_rocprofv3_kernel_trace bench_line_under_rocprofv3_kernel_trace_one_layout \
         bench_basin2048_f32_storage_study rehearsal_2ranks_one_gpu_basin1024 rehearsal_4ranks_one_gpu_basin1024 rehearsal_4ranks_one_gpu_basin2048 rehearsal_2ranks_one_gpu_basin1024_f32_storage \
         tile_probe_1tiles tile_probe_2tiles tile_probe_4tiles tile_probe_8tiles tile_probe_8tiles_first_tile tile_probe_8tiles_stand_in_mover tile_probe_8tiles_2x4 tile_probe_8tiles_same_box_as_2x4 tile_probe_8tiles_round30us_results_rounds tile_probe_8tiles_round30us_rim_rounds tile_probe_8tiles_rccl_self_RIM_MAIN tile_probe_8tiles_rccl_self_RIM_RESULTS_MAIN; do
  [ -s $O/$f.json ] && cp $O/$f.json $P/round${N}_$f.json || echo "missing $f"
done
cp "$(find $O/stats -name '*kernel_stats.csv' | head -1)" $P/round${N}_kernel_stats_basin2048.csv
cp "$(find $O/stats_one_layout -name '*kernel_stats.csv' | head -1)" $P/round${N}_kernel_stats_basin2048_one_layout.csv
cp $O/pmc_summary.csv $P/round${N}_pmc_fetch_write_basin2048.csv
cp $O/traffic.json $P/traffic.json
TUNED=$(python3 -c "import json; print(json.load(open('$O/bench_basin2048.json'))['ms_per_step'])")
python3 tools/make_tile_bounds.py --tuned-ms $TUNED $O/tile_probe_1tiles.json $O/tile_probe_2tiles.json $O/tile_probe_4tiles.json $O/tile_probe_8tiles.json > /dev/null
python3 tools/design_tables.py $P/round${N}_bench_basin2048_with_traffic.json
