#!/usr/bin/env python3
"""profiles/tile_probe_bounds.json from the lines tools/tile_probe.py printed in ONE job (same box, same library): what one tile of the
2 / 4 / 8-tile split of a bench workload costs per step outside the transfers, against the single tile of the same job -- the
strong-scaling efficiency a measured curve can at most reach.  bench.py puts it on its line (`efficiency_bound_from_tile_probe`).

    python tools/make_tile_bounds.py gpurun_out/<dir>/tile_1.json gpurun_out/<dir>/tile_2.json ...
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
tuned = None                                         # --tuned-ms X: ms per step of bench.py's N = 1 line (placement tuned) of the same round, to be read beside
if "--tuned-ms" in args:
    k = args.index("--tuned-ms")
    tuned = float(args[k + 1])
    args = args[:k] + args[k + 2:]
lines = [json.load(open(p)) for p in args]
out = {}
for wl in sorted({l["workload_key"] for l in lines}):
    mine = [l for l in lines if l["workload_key"] == wl]
    ntiles = lambda l: eval(l["tiles"].replace("x", "*"))
    one = [l for l in mine if ntiles(l) == 1]
    rec = {"library_build_id": mine[0]["library_build_id"], "what": "ms per step (wall) of ONE tile of the N-tile split on one MI355X through the multi-tile code path, "
           "stand-in mover (device copies on the round's stream, no link latency): tools/tile_probe.py", "tiles": {}}
    if one:
        rec["single_gpu_ms_per_step"] = one[0]["ms_per_step_wall"]
        rec["single_gpu_note"] = "the single tile through the same tool in the same job, its 3-D arrays where the allocator put them (like the tiles)"
    if tuned:
        rec["bench_n1_tuned_ms_per_step"] = tuned
    for l in sorted(mine, key=ntiles):
        n = ntiles(l)
        if n == 1:
            continue
        e = {"split": l["tiles"], "tile": l["tile"], "ms_per_step": l["ms_per_step_wall"], "message_rounds_between_kernels": l["message_rounds_per_step"],
             "message_rounds_on_second_stream": l["message_rounds_on_side_stream_per_step"]}
        if one:
            e["efficiency_bound"] = round(one[0]["ms_per_step_wall"] / (n * l["ms_per_step_wall"]), 3)
        if tuned:
            e["efficiency_bound_against_tuned_n1"] = round(tuned / (n * l["ms_per_step_wall"]), 3)
        rec["tiles"][str(n)] = e
    out[wl] = rec
with open(os.path.join(ROOT, "profiles", "tile_probe_bounds.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
