cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "16 16" "16 32" "16 64" "8 32" "4 32"; do set -- $cfg
ACEZ_POSE_TILE=$1 ACEZ_POSE_WB=$2 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/tr_$1_$2 -o trace -- python tools/pose_kernels_timing.py > /dev/null 2>&1
echo "tile $1 wb $2"; cut -c1-100 $(find /tmp/tr_$1_$2 -name "*kernel_stats.csv" | head -1) | grep "pose\|adamw_k\|loss_k\|step_begin" | head -8
done 2>&1 | tee gpurun_out/pose_kernels.log
ACEZ_LIB=$PWD/tools/libacez_r02.so python bench.py --headline-only --steps 100 --warmup 20 --buffer-patches 1000000 2>&1 | grep metric | cut -c1-200
