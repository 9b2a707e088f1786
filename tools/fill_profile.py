"""Where the time of one augmented buffer fill goes (session._fill_buffer_augmented: 8 M rows = 7813 views of 480 x 640 through warp -> encoder ->
sampling).  python tools/fill_profile.py [images] ; under rocprofv3 --kernel-trace --stats the kernel side, the wall clock printed here the whole."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
from acezero_amd import synth
from acezero_amd.session import ReconstructionSession, default_options
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seq = synth.render_room_sequence(seed=2089, n_frames=n, arc_deg=0.144 * n, device="cuda")
esd = {k: torch.from_numpy(v) for k, v in synth.init_encoder_weights_bandpass(seed=4099).items()}
opt = default_options(use_external_focal_length=seq["focal"], aug_rotation=2, aug_scale=1.06, aug_black_white=0.02)
ses = ReconstructionSession(esd, seq["images"], opt=opt, depth=seq["depth"])
ids = list(range(n))
poses = seq["poses"].cpu()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    buf = ses._fill_buffer_augmented(ids, poses, seq["focal"], False)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("fill %d: %d rows in %.3f s" % (rep, buf["features"].shape[0], dt), flush=True)
    del buf
# the phases, each alone on the same inputs (64 views of one scale level)
from acezero_amd.session import warp_views
from acezero_amd.buffer import BufferBuilder
b = 64
imgs = ses.images[torch.arange(b, device=ses.dev) % n]
ang = np.radians(np.random.default_rng(1).uniform(-2, 2, size=b))
def timed(f, reps=20):
    f(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.time() - t0) / reps * 1e3
t_warp = timed(lambda: warp_views(imgs, 1.0, ang, None))
views, masks, grid = warp_views(imgs, 1.0, ang, None)
t_enc = timed(lambda: ses.enc.features_rows(views))
bld = BufferBuilder(ses.enc, capacity=64 * 1024 * 24, samples_per_image=1024, seed=1)
eye = torch.eye(4).repeat(b, 1, 1); K = torch.eye(3).repeat(b, 1, 1)
def add():
    if bld.n + b * 1024 > bld.capacity: bld.n = 0; bld.n_views = 0
    bld.add_views(views, masks.float(), eye, eye, K, K, list(range(b)), check_empty=False)
t_add = timed(add)
print("per 64 views: warp %.2f ms, encoder alone %.2f ms, add_views (encoder + mask + sampling + tables) %.2f ms" % (t_warp, t_enc, t_add))
