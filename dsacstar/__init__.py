"""Drop-in for `import dsacstar` (register_mapping.py:12 of the reference): the reference builds this module from
dsacstar/dsacstar.cpp with OpenCV; here it is the MI355X implementation behind the same callable.

    import dsacstar
    inliers = dsacstar.forward_rgb(scene_coordinates_1x3xHxW, out_pose_4x4, hypotheses, threshold, focal, ppX, ppY,
                                   inlier_alpha, max_reproj, subsampling, seed, max_hypotheses_tries)

Put the repository root on PYTHONPATH (or copy this directory next to register_mapping.py) and the reference's
register_mapping.py runs unchanged on this call. Extras (not in the reference): register_batch (device-resident, batched),
set_verbose, reset_call_counter."""
from acezero_amd.dsacstar import forward_rgb, register_batch, reset_call_counter, set_verbose  # noqa: F401
