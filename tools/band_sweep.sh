#!/bin/bash
# developer tool: sweep the per-level L2 budget of a band (POMGPU_BAND_BYTES) on the default workload
for b in ${BANDS:-1048576 2097152 3145728 6291456 25165824}; do
  echo "BAND $b"
  POMGPU_BAND_BYTES=$b python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(round(d['ms_per_step'], 2), 'ms/step; internal', d['internal_mode']['device_ms_per_step'], 'ms;', d['kernel_time_share'])"
done
