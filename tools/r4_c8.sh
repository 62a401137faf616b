set -x
export TMPDIR=/tmp
PMC_CMD="tools/tile_probe.py --tiles 8 --rank 4 --steps 1" bash tools/pmc_kernel.sh tile8 'k_advt2x2|k_advq2|k_advuv|k_advct_col|k_profq|k_ts_update|k_uv_filter|k_ext_march2|k_proft'
