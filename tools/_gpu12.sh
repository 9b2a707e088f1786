cd $GRAFT_REPO_ROOT
for cfg in "16 4" "32 4" "16 8" "32 8"; do set -- $cfg
ACEZ_POSE_WB=$1 ACEZ_POSE_WW=$2 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/trx_$1_$2 -o trace -- python bench.py --headline-only --steps 100 --warmup 30 --buffer-patches 1000000 --pose-refinement mlp > /tmp/o_$1_$2.log 2>&1
echo "fused wb $1 ww $2: $(grep -o '"ms_per_step": [0-9.]*' /tmp/o_$1_$2.log)"; cut -c1-100 $(find /tmp/trx_$1_$2 -name "*kernel_stats.csv" | head -1) | grep "pose_mlp_wgrad"
ACEZ_POSE_WB=$1 ACEZ_POSE_WW=$2 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/try_$1_$2 -o trace -- python tools/pose_kernels_timing.py > /dev/null 2>&1
echo "split wb $1 ww $2"; cut -c1-100 $(find /tmp/try_$1_$2 -name "*kernel_stats.csv" | head -1) | grep "pose_mlp_wgrad"
done 2>&1 | tee gpurun_out/wgrad_variants.log
