import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_chain_gpu import _big_problem
from acezero_amd.head import HeadTrainer
from acezero_amd import synth
def P(*a):
    print(*a, flush=True)
prob = _big_problem(n_images=8, patches_per_view=256)
n = 1024
for nb, homog in ((2, True), (0, True), (1, False)):
    trs = []
    for seq in ("0", "1"):
        os.environ["ACEZ_SEQ"] = seq
        tr = HeadTrainer(prob["mean"], num_head_blocks=nb, use_homogeneous=homog, max_batch=n, loss_type="tanh", schedule="constant", iterations=50, lr_min=3e-4)
        tr.load_flat(torch.from_numpy(synth.init_head_params(11, num_head_blocks=nb, use_homogeneous=homog)))
        tr.set_buffer(prob["features"], prob["target_px"], prob["view_idx"], prob["view_aug_inv"], prob["view_K"], prob["view_Kinv"], prob["view_image"], prob["image_pose_inv"])
        trs.append(tr)
    ref, new = trs
    rng = np.random.default_rng(3)
    for it in range(5):
        idx = torch.from_numpy(rng.permutation(prob["features"].shape[0])[:n].astype(np.int64)).cuda()
        ref.step(idx); torch.cuda.synchronize(); P("ref step", nb, it)
        new.step(idx); torch.cuda.synchronize(); P("new step", nb, it)
    P("params equal", torch.equal(ref.params, new.params))
    for rows in (1, 333, 1024, 3000):
        f = torch.from_numpy(prob["features"][:rows]).cuda()
        a = ref.get_scene_coordinates(f); torch.cuda.synchronize(); P("ref infer", rows)
        b = new.get_scene_coordinates(f); torch.cuda.synchronize(); P("new infer", rows, torch.equal(a, b))
