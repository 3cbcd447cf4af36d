"""Seeded synthetic inputs of the shapes the pipeline sees (the reference's datasets are not
redistributable and absent): camera, landmark clouds, 2D-3D correspondences with outliers."""
import numpy as np

# KITTI 05 camera matrix, main_demo.m:58-60
K_KITTI = np.array([[718.856, 0.0, 607.1928],
                    [0.0, 718.856, 185.2157],
                    [0.0, 0.0, 1.0]])
KITTI_W, KITTI_H = 1241, 376


def rotation_from_rotvec(w):
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Wx
    return np.eye(3) + np.sin(th) / th * Wx + (1 - np.cos(th)) / th ** 2 * Wx @ Wx


def pnp_problem(n, inlier_ratio=0.7, noise_px=0.5, seed=0, K=K_KITTI, width=KITTI_W, height=KITTI_H,
                depth=(5.0, 80.0)):
    """2D-3D correspondences for ransacLocalization: returns kp_rc (2xN, (row, col)), X (3xN),
    R_C_W, t_C_W (ground truth) and the boolean inlier labels."""
    rng = np.random.default_rng(seed)
    R_C_W = rotation_from_rotvec(rng.normal(0, 0.05, 3))
    t_C_W = rng.normal(0, 0.5, 3)
    u = rng.uniform(0, width, n)
    v = rng.uniform(0, height, n)
    z = rng.uniform(depth[0], depth[1], n)
    pc = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z])
    X = R_C_W.T @ (pc - t_C_W[:, None])
    is_in = rng.random(n) < inlier_ratio
    u_obs = np.where(is_in, u + rng.normal(0, noise_px, n), rng.uniform(0, width, n))
    v_obs = np.where(is_in, v + rng.normal(0, noise_px, n), rng.uniform(0, height, n))
    kp_rc = np.stack([v_obs, u_obs])
    return kp_rc, X, R_C_W, t_C_W, is_in


def sample_triples(n, count, seed=42):
    """`count` triples of distinct indices in [0, n) from numpy's PCG64(seed)."""
    from .ransac import draw_triples
    return draw_triples(np.random.Generator(np.random.PCG64(seed)), n, count)
