#!/usr/bin/env python3
"""developer tool: the per-kernel table of DESIGN.md section 4 from a bench line (kernel_ms_per_step of its profiled step), bench.py's
algorithmic pass counts and profiles/traffic.json (HBM-side bytes per launch from the PMC passes of the same build).

    python tools/design_tables.py profiles/round5_bench_basin2048_with_traffic.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ROWS = {   # kernel as the profile names it -> (reference lines, name in traffic.json)
    "k_profq": ("a10 + a11", "`solver.f:1212-1538`, `advance.f:416-421`", "k_profq"),
    "k_ts_update": ("a15 + a16 + restore", "`advance.f:444-454`, `solver.f:1162-1209`", "k_ts_update"),
    "k_advt2x2_col": ("a13 x 2", "`solver.f:577-731`", "k_advt2_col"),
    "k_advuv_col": ("a17", "`solver.f:734-845`", "k_advuv_col"),
    "k_advq2_col": ("a9 x 2", "`solver.f:411-477`", "k_advq_col"),
    "k_advct_col": ("a2", "`solver.f:201-408`", "k_advct_col"),
    "k_uv_filter_reg2": ("a19", "`advance.f:469-514`", "k_uv_filter_reg"),
    "k_profuv_reg2": ("a18", "`solver.f:1686-1877`", "k_profuv_reg"),
    "k_proft_reg2": ("a14 x 2", "`solver.f:1541-1683`", "k_proft_reg"),
    "k_baropg": ("a3", "`solver.f:848-940`", "k_baropg_rs"),
    "k_int_uvmean_reg2": ("a7", "`advance.f:365-393`", "k_int_uvmean_reg"),
    "k_realvertvl_col": ("a20", "`solver.f:2024-2067`", "k_realvertvl_col"),
    "k_vertvl": ("a8", "`solver.f:1970-2021`", "k_vertvl_rs"),
    "k_aam_pair": ("a1", "`advance.f:122-137`", "k_aam_pair"),
}
PASSES = dict(bench.KERNEL_PASSES, k_uv_filter_reg2=10, k_profuv_reg2=6, k_proft_reg2=6, k_int_uvmean_reg2=4, k_baropg=4, k_vertvl=3)
TWIN = {"k_uv_filter_reg2", "k_profuv_reg2", "k_proft_reg2", "k_int_uvmean_reg2"}   # traffic.json holds one launch of the pair (older name): x 2

line = json.load(open(sys.argv[1]))
ms = line["kernel_ms_per_step"]
cells = line["config"]["global_cells"]
tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
same = tj.get("_build_id") == line.get("library_build_id")
print("| kernel | rows | reference | passes | ms | of 8 TB/s | HBM-side / algorithmic |")
print("|---|---|---|---|---|---|---|")
for k, (rows, ref, tname) in ROWS.items():
    if k not in ms:
        continue
    p = PASSES[k]
    alg = p * 8.0 * cells
    frac = alg / (ms[k] * 1e-3) / 1e9 / bench.HBM_PEAK_GBS
    rec = tj.get(f"basin2048/1/{k}") or tj.get(f"basin2048/1/{tname}")
    ratio = ""
    if rec:
        real = rec["bytes_per_launch"] * (2 if (k in TWIN and f"basin2048/1/{k}" not in tj) else 1)
        ratio = f"{real / alg:.2f}" + ("" if same else " (earlier build)")
    print(f"| `{k}` | {rows} | {ref} | {p} | {ms[k]:.2f} | {frac:.2f} | {ratio} |")
ext = sum(v for n, v in ms.items() if n.startswith(("k_ext_", "k_advave_")))
print(f"| `k_ext_pair` + `k_ext_ring` + `k_advave_pair` | a5 + a6 + a22 | `advance.f:205-353`, `solver.f:6-198` | 2-D | {ext:.2f} | -- | -- |")
im = line["internal_mode"]
print()
print(f"step {line['ms_per_step']:.2f} ms = {line['value']:.3e} cell-updates/s; internal mode {im['device_ms_per_step']} ms = {im['frac_of_peak']} of 8 TB/s "
      f"({im['frac_of_measured_copy_ceiling']} of the copy ceiling); external mode {line['external_mode']['device_ms_per_step']} ms; "
      f"untuned layout of that box {line.get('n1_untuned_ms_per_step')} ms; roofline {line['roofline']}")
