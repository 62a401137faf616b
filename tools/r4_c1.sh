set -x
export TMPDIR=/tmp
O=gpurun_out/r4c1; mkdir -p $O
timeout -k 10 120 ./tools/micro/grid_barrier > $O/grid_barrier.txt 2>&1; tail -12 $O/grid_barrier.txt
for n in 8 4 2; do timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_$n.json 2> $O/tile_$n.err; tail -c 300 $O/tile_$n.json; done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench1.json 2> $O/bench1.err; echo "bench rc=$?"; tail -c 600 $O/bench1.json
POM_BENCH_REHEARSE=1 timeout -k 10 500 python3 bench.py --gpus 2 --workload basin1024 --steps 3 --warmup 1 > $O/rehearse2.json 2> $O/rehearse2.err; echo "rehearse2 rc=$?"; tail -c 1500 $O/rehearse2.json
