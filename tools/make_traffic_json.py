"""developer tool: profiles/traffic.json from a pmc_summary.csv (tools/pmc_summarise.py output holding
FETCH_SIZE and WRITE_SIZE columns).  bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- see the
note written into the file.  The file names the library build its counters were taken from (pomgpu_build_id of the
libpomgpu.so in the tree -- run this on the snapshot that was profiled); bench.py quotes roofline.traffic only for that build.
usage: make_traffic_json.py pmc_summary.csv [workload] [n_gpus]"""
import csv, json, os, sys
src = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "basin2048"
world = sys.argv[3] if len(sys.argv) > 3 else "1"
out = {"_note": "HBM-side traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, 1 GPU): "
       "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. The factor 2 on FETCH_SIZE is MI355X_MICROARCH.md's gfx950 correction, "
       "re-calibrated on this code's own 8-byte-per-lane loads: k_roundtrip reads two 1.258 GB arrays and reports "
       "FETCH_SIZE = 1.229e6 KiB (= one array), k_q_filter reads six and reports three; WRITE_SIZE is exact (k_roundtrip: "
       "1.229e6 KiB for one array). Infinity-Cache hits are included (L2-miss traffic)."}
for r in csv.DictReader(open(src)):
    if not r.get("FETCH_SIZE") or not r.get("WRITE_SIZE"):
        continue
    f, w = float(r["FETCH_SIZE"]), float(r["WRITE_SIZE"])
    out[f"{workload}/{world}/{r['kernel']}"] = {"bytes_per_launch": int((2 * f + w) * 1024), "fetch_size_kib": f, "write_size_kib": w}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from extpom_amd import lib as _lib
out["_build_id"] = _lib.load().pomgpu_build_id().decode()
json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print(len(out) - 2, "kernels; build", out["_build_id"])
