# developer A/B: the balanced XCD order of the row-sharing kernels on tiles (POMGPU_NO_LIN=1 = the banded order)
for spec in 8:1x8:4 8:4x2:5 4:1x4:1 2:1x2:0; do
  IFS=: read n g r <<< "$spec"
  for lin in 0 1; do
    if [ $lin = 0 ]; then export POMGPU_NO_LIN=1; else unset POMGPU_NO_LIN; fi
    POM_TILE_GRID=$g timeout -k 10 120 python tools/tile_probe.py --tiles $n --rank $r > gpurun_out/lin_${n}_${g}_$lin.json 2> gpurun_out/lin_${n}_${g}_$lin.err || { echo FAIL $spec; tail -3 gpurun_out/lin_${n}_${g}_$lin.err; }
    python - <<P
import json
d=json.load(open("gpurun_out/lin_${n}_${g}_$lin.json"))
k=d["kernels"]
print("$spec lin=$lin", d["tile"], "wall", d["ms_per_step_wall"], " ".join(f"{n[2:]} {k[n][1]}" for n in ("k_advt2x2_col","k_advq2_col","k_advuv_col","k_advct_col","k_ts_update","k_profq")), flush=True)
P
  done
done
