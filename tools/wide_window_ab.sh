# developer A/B: shrinking row window of the extended tile's external substeps (POMGPU_WIDE_FULL=1 = every row, every substep)
for spec in 8:1x8:4 8:4x2:5 4:1x4:1; do
  IFS=: read n g r <<< "$spec"
  for full in 1 0; do
    if [ $full = 1 ]; then export POMGPU_WIDE_FULL=1; else unset POMGPU_WIDE_FULL; fi
    POM_TILE_GRID=$g timeout -k 10 120 python tools/tile_probe.py --tiles $n --rank $r > gpurun_out/ww_${n}_${g}_$full.json 2> gpurun_out/ww_${n}_${g}_$full.err || { echo FAIL $spec; tail -3 gpurun_out/ww_${n}_${g}_$full.err; }
    python - <<P
import json
d=json.load(open("gpurun_out/ww_${n}_${g}_$full.json"))
k=d["kernels"]
print("$spec full=$full", d["tile"], "wall", d["ms_per_step_wall"], " ".join(f"{n[2:]} {k[n][1]}" for n in k if n.startswith("k_ext")), flush=True)
P
  done
done
