for spec in 8:4x2:5 8:2x4:3 8:1x8:4 8:8x1:4 4:2x2:1 4:1x4:1 4:4x1:1 2:2x1:0 2:1x2:0; do
  IFS=: read n g r <<< "$spec"
  POM_TILE_GRID=$g timeout -k 10 120 python tools/tile_probe.py --tiles $n --rank $r > gpurun_out/grid_${n}_${g}.json 2> gpurun_out/grid_${n}_${g}.err || { echo FAIL $spec; tail -3 gpurun_out/grid_${n}_${g}.err; }
  python - <<P
import json
d=json.load(open("gpurun_out/grid_${n}_${g}.json"))
k=d["kernels"]
ext=sum(v[1] for n,v in k.items() if n.startswith(("k_ext_","k_advave","k_modeint_tail","k_int_tail","k_check","k_copy2")))
print("$spec", d["tile"], "wall", d["ms_per_step_wall"], "ext", round(ext,3), "profq", k["k_profq"][1], "ts", k["k_ts_update"][1], "copy+pack", round(sum(v[1] for n,v in k.items() if n.startswith(("k_rect_copy","k_halo_"))),3), "rounds", d["message_rounds_per_step"], flush=True)
P
done
