"""Pins the ICP restatement (oracle/genpc_oracle_geom.c): Horn's quaternion solve
against numpy's SVD-based Kabsch with the reflection fix (what Eigen::umeyama
computes), and the loop by recovering known rigid motions.  open3d itself is
absent and unpinned in the reference: parity unpinned for this row."""
import math

import numpy as np
import pytest


def np_kabsch(P, Q):
    mp, mq = P.mean(0), Q.mean(0)
    H = (P - mp).T @ (Q - mq)
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, mq - R @ mp


def sums_of(P, Q, d2=None):
    s = np.zeros(17)
    s[0] = len(P)
    s[1:4] = P.sum(0)
    s[4:7] = Q.sum(0)
    s[7:16] = (P.T @ Q).reshape(9)
    s[16] = 0 if d2 is None else d2.sum()
    return s


def rot(axis, deg):
    a = np.asarray(axis, float)
    a /= np.linalg.norm(a)
    th = math.radians(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_kabsch_matches_svd(oracle, seed):
    rng = np.random.default_rng(seed)
    P = rng.standard_normal((200, 3))
    if seed == 3:
        P[:, 2] = 0.0                               # coplanar source
    R0 = rot(rng.standard_normal(3), rng.uniform(-170, 170))
    Q = P @ R0.T + rng.standard_normal(3) + 0.01 * rng.standard_normal((200, 3))
    U = oracle.kabsch_from_sums(sums_of(P, Q))
    R, t = np_kabsch(P, Q)
    np.testing.assert_allclose(U[:3, :3], R, atol=1e-9)
    np.testing.assert_allclose(U[:3, 3], t, atol=1e-9)
    np.testing.assert_allclose(U[3], [0, 0, 0, 1])


def test_icp_recovers_rigid_motion(oracle):
    rng = np.random.default_rng(5)
    u = rng.standard_normal((3000, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    target = (u * np.array([0.5, 0.3, 0.2]) + 0.05 * np.abs(u[:, :1])).astype(np.float32)
    R0, t0 = rot([0.2, 1, 0.1], 6.0), np.array([0.02, -0.015, 0.01])
    src = ((target[::2].astype(np.float64) - t0) @ R0).astype(np.float32)      # target = R0 src + t0
    T, fit, rmse, its = oracle.icp(src, target, 0.075)
    assert fit == 1.0 and rmse < 2e-3 and 1 <= its <= 30
    np.testing.assert_allclose(T[:3, :3], R0, atol=5e-3)
    np.testing.assert_allclose(T[:3, 3], t0, atol=2e-3)
    # a tight correspondence threshold leaves points without a match: fitness < 1
    T2, fit2, _, _ = oracle.icp(src, target, 0.01)
    assert 0 < fit2 < 1
    # init is honoured and composed on the left
    T3, _, _, its3 = oracle.icp(src, target, 0.075, init=T)
    assert its3 <= 2
    np.testing.assert_allclose(T3, T, atol=1e-4)
