"""CPU: the two roma 1.4.1 functions the pose refiner calls (refine_poses.py:136-150) are not vendored in the reference and are
restated in oracle/head_oracle.py ("unpinned"). These tests pin them on their mathematical DEFINITIONS with
independent tools: special_procrustes = the rotation nearest to M in the Frobenius norm (checked against scipy's orthogonal
Procrustes solver, against a brute-force search around the result and against the first-order optimality condition), and
special_gramschmidt = QR-style orthonormalisation of the first two columns with the right-handed third (checked against
numpy's QR with sign fix). Properties roma documents: outputs are rotations (orthonormal, det +1) and rotations are fixed points."""
import numpy as np
import pytest
import torch
from scipy.linalg import orthogonal_procrustes
from scipy.spatial.transform import Rotation

from oracle import head_oracle


def _mats(seed, n, noise):
    rng = np.random.default_rng(seed)
    R = Rotation.random(n, random_state=seed).as_matrix()
    return R + noise * rng.normal(size=(n, 3, 3))


@pytest.mark.parametrize("noise", [0.0, 0.05, 0.6])
def test_special_procrustes_is_the_nearest_rotation(noise):
    M = _mats(3, 40, noise)
    R = head_oracle.special_procrustes(torch.from_numpy(M)).numpy()
    for k in range(len(M)):
        assert np.allclose(R[k] @ R[k].T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R[k]) - 1) < 1e-12
        # scipy: argmin_Q ||I Q - M||_F over orthogonal Q (here det(M) > 0 for small noise, so Q is the nearest rotation)
        Q, _ = orthogonal_procrustes(np.eye(3), M[k])
        if np.linalg.det(Q) > 0:
            assert np.allclose(R[k], Q, atol=1e-10)
        # optimality: no rotation in a neighbourhood is closer, and R^T M is symmetric (first-order condition)
        S = R[k].T @ M[k]
        assert np.allclose(S, S.T, atol=1e-10)
        d0 = np.linalg.norm(R[k] - M[k])
        for dR in Rotation.from_rotvec(1e-3 * np.random.default_rng(k).normal(size=(20, 3))).as_matrix():
            assert np.linalg.norm(R[k] @ dR - M[k]) >= d0 - 1e-12
    if noise == 0.0:
        assert np.allclose(R, M, atol=1e-12)                           # rotations are fixed points


def test_special_procrustes_reflection_case():
    """det(M) < 0: the nearest ROTATION flips the smallest singular direction (the diag(1, 1, det(U V^T)) factor)."""
    M = _mats(5, 10, 0.1) @ np.diag([1.0, 1.0, -1.0])
    R = head_oracle.special_procrustes(torch.from_numpy(M)).numpy()
    for k in range(len(M)):
        assert abs(np.linalg.det(R[k]) - 1) < 1e-12
        d0 = np.linalg.norm(R[k] - M[k])
        for Q in Rotation.random(200, random_state=k).as_matrix():
            assert np.linalg.norm(Q - M[k]) >= d0 - 1e-12              # no rotation at all is closer


@pytest.mark.parametrize("noise", [0.0, 0.05, 0.6])
def test_special_gramschmidt_matches_qr(noise):
    M = _mats(7, 40, noise)
    G = head_oracle.special_gramschmidt(torch.from_numpy(M)).numpy()
    for k in range(len(M)):
        Q, Rr = np.linalg.qr(M[k][:, :2])
        Q = Q * np.sign(np.diag(Rr))                                    # Gram-Schmidt keeps the columns' own directions
        assert np.allclose(G[k][:, :2], Q, atol=1e-12)
        assert np.allclose(G[k][:, 2], np.cross(Q[:, 0], Q[:, 1]), atol=1e-12)
        assert np.allclose(G[k] @ G[k].T, np.eye(3), atol=1e-12) and abs(np.linalg.det(G[k]) - 1) < 1e-12
    if noise == 0.0:
        assert np.allclose(G, M, atol=1e-12)
