set -x
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
./tools/micro/grid_barrier > $O/grid_barrier.txt 2>&1; tail -12 $O/grid_barrier.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -5 $O/gputests.log
for n in 8 4 2; do timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_$n.json 2> $O/tile_$n.err; tail -c 300 $O/tile_$n.json; done
POM_BENCH_REHEARSE=1 timeout -k 10 500 python3 bench.py --gpus 2 --workload basin1024 --steps 3 --warmup 1 > $O/rehearse2.json 2> $O/rehearse2.err; echo "rehearse2 rc=$?"; tail -c 1500 $O/rehearse2.json
