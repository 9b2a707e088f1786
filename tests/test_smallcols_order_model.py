"""CPU: the summation ORDER of the column-parallel small-parameter reduction (acezero_amd/csrc/head_kernels.hip: SmallCols<4> / <8>,
used by adamw_small_columns, grad_reduce_kernel's tail and the multiplier waves of wgrad_opt_kernel) is the order of tail_output (one
wavefront per output: lane j adds the partial rows j, j + 64, ... in that order, then the xor butterfly 32, 16, ..., 1), so the two
produce the same float32 bits. The kernels are compared on the GPU (tests/test_head_gpu.py, tests/test_wgrad_opt_gpu.py); this file
pins the argument: with LPO lanes per output, lane q holds the partials j = LPO i + q, the butterfly levels >= LPO pair i with
i ^ (off / LPO) inside the lane, the remaining levels pair lanes q and q ^ off -- the same binary tree, float addition being commutative."""
import numpy as np
import pytest

f32 = np.float32


def tail_output(rows):
    """rows: float32 [cnt] partial rows of one output. A wavefront of 64 lanes, as the kernel does it."""
    acc = np.zeros(64, f32)
    for j in range(64):
        for b in range(j, len(rows), 64):
            acc[j] = f32(acc[j] + rows[b])
    off = 32
    while off >= 1:
        acc = np.array([f32(acc[j] + acc[j ^ off]) for j in range(64)], f32)   # every lane at once (__shfl_xor)
        off >>= 1
    assert len(set(acc.view(np.uint32).tolist())) == 1      # all lanes agree
    return acc[0]


def small_cols(rows, lpo):
    ni = 64 // lpo
    cnt = len(rows)
    lane_val = []
    for q in range(lpo):
        acc = np.zeros(ni, f32)
        u = 0
        while 64 * u < cnt:                                   # rounds of 64 rows, in order (the first five are requested at once)
            for i in range(ni):
                row = 64 * u + lpo * i + q
                acc[i] = f32(acc[i] + (rows[row] if row < cnt else f32(0)))
            u += 1
        off = ni // 2
        while off >= 1:
            for i in range(off):
                acc[i] = f32(acc[i] + acc[i + off])
            off >>= 1
        lane_val.append(acc[0])
    g = np.array(lane_val, f32)
    off = lpo // 2
    while off >= 1:                                           # __shfl_xor(4) / DPP quad_perm [2,3,0,1] / [1,0,3,2]
        g = np.array([f32(g[q] + g[q ^ off]) for q in range(lpo)], f32)
        off >>= 1
    assert len(set(g.view(np.uint32).tolist())) == 1
    return g[0]


@pytest.mark.parametrize("cnt", [1, 5, 41, 64, 65, 100, 160, 320, 333, 640])
@pytest.mark.parametrize("lpo", [4, 8])
def test_same_bits_as_the_wavefront_per_output_reduction(cnt, lpo):
    rng = np.random.default_rng(cnt * 10 + lpo)
    for scale in (1.0, 1e-3, 1e4):
        rows = (rng.standard_normal(cnt) * scale).astype(f32)
        rows[rng.integers(0, cnt)] = f32(0)
        a, b = tail_output(rows), small_cols(rows, lpo)
        assert a.view(np.uint32) == b.view(np.uint32), (cnt, lpo, a, b)
