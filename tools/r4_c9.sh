set -x
export TMPDIR=/tmp
O=gpurun_out/r4c9; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=6 > $O/gputests.log 2>&1; echo "gputests rc=$?"; tail -14 $O/gputests.log
