#!/bin/bash
# developer tool (GPU box): HBM-side traffic and L2 counters of selected kernels for one library variant
# usage: tools/pmc_kernel.sh <tag> <kernel-regex> [lib]        (PMC_CMD="tools/tile_probe.py --tiles 8 --rank 4 --steps 1": another program than the bench)
export TMPDIR=/tmp
TAG=$1; PAT=$2; LIB=$3
O=gpurun_out/pmc_$TAG; mkdir -p $O
[ -n "$LIB" ] && export POMGPU_LIBPATH=$PWD/$LIB
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python3 ${PMC_CMD:-bench.py --steps 1 --warmup 1 --no-cpu-baseline} > $O/$tag.log 2>&1
done
python3 tools/pmc_summarise.py $O > $O/summary.csv 2>&1
python3 - "$O/summary.csv" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if not re.search(sys.argv[2], r["kernel"]): continue
    f = lambda k: float(r[k]) if r.get(k) else 0.0
    cyc = f("GRBM_GUI_ACTIVE") / 8 or 1
    print(f"{r['kernel']:20s} FETCHx2+WRITE={(2*f('FETCH_SIZE')+f('WRITE_SIZE'))*1024/1e9:.2f} GB (F={2*f('FETCH_SIZE')*1024/1e9:.2f} W={f('WRITE_SIZE')*1024/1e9:.2f})  L2hit={f('TCC_HIT_sum')/max(f('TCC_HIT_sum')+f('TCC_MISS_sum'),1):.2f} TCC_REQ={f('TCC_REQ_sum'):.3e} TCP->TCC rd={f('TCP_TCC_READ_REQ_sum'):.3e} TCPacc/CUcyc={f('TCP_TOTAL_CACHE_ACCESSES_sum')/(cyc*256):.2f} waitany={f('SQ_WAIT_ANY')/max(f('SQ_WAVE_CYCLES'),1):.2f} valu/wavecyc={f('SQ_ACTIVE_INST_VALU')/max(f('SQ_WAVE_CYCLES'),1):.2f} cyc={cyc:.3e}")
PY
