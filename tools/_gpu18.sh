cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_head_fp16_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|where\|built-in" | tail -40 > gpurun_out/t1.log; cat gpurun_out/t1.log
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_seq_gpu.py tests/test_pose_fused_gpu.py tests/test_chain_gpu.py tests/test_dp_gpu.py -m gpu -x -q --tb=short 2>&1 | grep -v "where\|built-in" | tail -15 > gpurun_out/t2.log; cat gpurun_out/t2.log
ACEZ_LIB=$PWD/tools/libacez_r02.so python tools/_diag_fp16.py bf16 /tmp/p_r02.npy 2>&1 | tail -1
python tools/_diag_fp16.py bf16 /tmp/p_new.npy 2>&1 | tail -1
python -c "
import numpy as np
a=np.load('/tmp/p_r02.npy'); b=np.load('/tmp/p_new.npy')
print('bf16 split flow, one step, this build vs round 2: bitwise equal', np.array_equal(a,b), int((a!=b).sum()))"
