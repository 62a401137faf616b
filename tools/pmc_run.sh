#!/bin/bash
# developer tool: per-kernel PMC counters of a short bench run (separate passes; no trace domains mixed in)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$1; shift
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
ARGS="--workload ${WL:-basin2048} --steps 1 --warmup 1 --no-cpu-baseline"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 bench.py $ARGS > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"
done
python3 tools/pmc_summarise.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
