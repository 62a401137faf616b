#!/bin/bash
# developer tool (GPU box): address-translation (UTCL1) counters per kernel for one bench step
# usage: tools/pmc_tlb.sh <tag>
export TMPDIR=/tmp
O=gpurun_out/tlb_$1; mkdir -p $O
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_GUI_ACTIVE" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/$tag.log 2>&1
done
python3 tools/pmc_summarise.py $O > $O/summary.csv 2>&1
