"""Seeded synthetic inputs of the shapes the pipeline sees (the reference's datasets are not
redistributable and absent): camera, landmark clouds, 2D-3D correspondences with outliers."""
import numpy as np

# KITTI 05 camera matrix, main_demo.m:58-60
K_KITTI = np.array([[718.856, 0.0, 607.1928],
                    [0.0, 718.856, 185.2157],
                    [0.0, 0.0, 1.0]])
KITTI_W, KITTI_H = 1241, 376
# The parking calibration the reference holds, data/epfl_parking/calibration/K.txt:1-3 (1280 x 960 frames,
# data/epfl_parking/readme.md); the K.txt of the 640 x 480 "parking" set (main_demo.m:100-105) is not in the
# repository, so synthetic parking sequences use this camera at half resolution.
K_EPFL_PARKING = np.array([[973.223300, 0.0, 656.959936],
                           [0.0, 971.330996, 488.683233],
                           [0.0, 0.0, 1.0]])
K_PARKING = np.array([[0.5, 0.0, 0.0], [0.0, 0.5, 0.0], [0.0, 0.0, 1.0]]) @ K_EPFL_PARKING
PARKING_W, PARKING_H = 640, 480
# Malaga 07 (main_demo.m:79-97, ds = 1): rectified 800 x 600 images
K_MALAGA = np.array([[621.18428, 0.0, 404.0076], [0.0, 621.18428, 309.05989], [0.0, 0.0, 1.0]])
MALAGA_W, MALAGA_H = 800, 600


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


def pose_matrix(rotvec, t):
    """4x4 T_W_C from a rotation vector and a camera position."""
    T = np.eye(4)
    T[:3, :3] = rotation_from_rotvec(rotvec)
    T[:3, 3] = t
    return T


def projection_matrix(K, T_W_C):
    """3x4 M = K * inv(T_W_C)(1:3, :)."""
    return K @ np.linalg.inv(T_W_C)[:3, :]


def two_view_problem(n, noise_px=0.3, seed=0, K=K_KITTI, baseline=2.0):
    """n world points seen from two cameras: returns p1, p2 (3xn homogeneous pixels), M1, M2, X (3xn)."""
    rng = np.random.default_rng(seed)
    T1 = pose_matrix(rng.normal(0, 0.02, 3), rng.normal(0, 0.1, 3))
    T2 = pose_matrix(rng.normal(0, 0.05, 3), np.array([baseline, 0.2, 0.5]) + rng.normal(0, 0.1, 3))
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 3, n), rng.uniform(8, 40, n)])
    M1, M2 = projection_matrix(K, T1), projection_matrix(K, T2)
    Xh = np.vstack([X, np.ones(n)])
    p1 = M1 @ Xh
    p1 /= p1[2]
    p2 = M2 @ Xh
    p2 /= p2[2]
    p1[:2] += rng.normal(0, noise_px, (2, n))
    p2[:2] += rng.normal(0, noise_px, (2, n))
    return p1, p2, M1, M2, X


def candidate_problem(n, seed=0, K=K_KITTI, noise_px=0.3):
    """State of update_landmarks after tracking: n candidates first seen from one of 6 earlier
    poses, now seen from the current pose.  Returns cur_rc, first_rc (2xn), T_first (12xn),
    T_W_C (4x4), X_true (3xn).  Depths 3..150 m and baselines 0..5 m so that every status occurs."""
    rng = np.random.default_rng(seed)
    T_W_C = pose_matrix(rng.normal(0, 0.03, 3), np.array([0.3, -0.1, 6.0]))
    firsts = [pose_matrix(rng.normal(0, 0.03, 3), np.array([rng.normal(0, 0.2), rng.normal(0, 0.05), z]))
              for z in (0.0, 1.0, 2.5, 4.0, 5.0, 5.9)]
    which = rng.integers(0, len(firsts), n)
    Mc = projection_matrix(K, T_W_C)
    u = rng.uniform(50, KITTI_W - 50, n)
    v = rng.uniform(20, KITTI_H - 20, n)
    z = rng.uniform(3, 150, n)
    pc = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z, np.ones(n)])
    X = (T_W_C @ pc)[:3]
    Xh = np.vstack([X, np.ones(n)])
    cur = Mc @ Xh
    cur = cur[:2] / cur[2] + rng.normal(0, noise_px, (2, n))
    first = np.zeros((2, n))
    T_first = np.zeros((12, n))
    for i in range(n):
        Tf = firsts[which[i]]
        pf = projection_matrix(K, Tf) @ Xh[:, i]
        first[:, i] = pf[:2] / pf[2] + rng.normal(0, noise_px, 2)
        T_first[:9, i] = Tf[:3, :3].T.reshape(9)  # column-major R
        T_first[9:, i] = Tf[:3, 3]
    return cur[::-1].copy(), first[::-1].copy(), T_first, T_W_C, X


def textured_image(height, width, seed=0, sigma=1.5, octaves=5):
    """Natural-image-like random texture as a float64 array (height, width) in 0..255: a sum of
    band-limited white-noise layers with Gaussian sigma, 2 sigma, 4 sigma, ... and equal
    variance per octave (a 1/f-like spectrum, so every pyramid level has trackable structure).
    Periodic, so it can be shifted exactly in the Fourier domain."""
    rng = np.random.default_rng(seed)
    fy = np.fft.fftfreq(height)[:, None]
    fx = np.fft.fftfreq(width)[None, :]
    img = np.zeros((height, width))
    for o in range(octaves):
        s = sigma * 2 ** o
        gauss = np.exp(-2.0 * (np.pi * s) ** 2 * (fx ** 2 + fy ** 2))
        layer = np.fft.ifft2(np.fft.fft2(rng.normal(size=(height, width))) * gauss).real
        img += layer / layer.std()
    lo, hi = np.percentile(img, [0.5, 99.5])
    return 255.0 * np.clip((img - lo) / (hi - lo), 0.0, 1.0)


def shift_image(img, dx, dy):
    """Exact (Fourier) sub-pixel shift of a periodic float image: out(y, x) = img(y - dy, x - dx)."""
    h, w = img.shape
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.fftfreq(w)[None, :]
    return np.fft.ifft2(np.fft.fft2(img) * np.exp(-2j * np.pi * (fx * dx + fy * dy))).real


def to_uint8(img):
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def twist_of(T):
    """[v; w] of a 4x4 rigid transform (numpy, via the closed-form log)."""
    R, t = T[:3, :3], T[:3, 3]
    cs = (np.trace(R) - 1.0) / 2.0
    s = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    sn = np.linalg.norm(s)
    th = np.arctan2(sn, cs)
    w = s * (th / sn) if sn > 1e-12 else s
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    th2 = th * th
    if th2 < 1e-10:
        Vinv = np.eye(3) - 0.5 * Wx + Wx @ Wx / 12.0
    else:
        A, B = np.sin(th) / th, (1 - np.cos(th)) / th2
        Vinv = np.eye(3) - 0.5 * Wx + (1 - A / (2 * B)) / th2 * Wx @ Wx
    return np.concatenate([Vinv @ t, w])


def ba_problem(F=5, m=300, visible=0.6, noise_px=0.5, rot_perturb=1e-2, trans_perturb=5e-2, lm_perturb=0.2,
               seed=9, K=K_KITTI, width=KITTI_W, height=KITTI_H):
    """A sliding-window BA problem in the reference's serialisation (VO_bundle_adjust.m:43-50):
    returns state0 (6F+3m, perturbed), obs ([F; m; per frame n_i; (row, col) pairs; 1-based ids]),
    state_true."""
    rng = np.random.default_rng(seed)
    poses = [pose_matrix(np.array([0.0, 0.01 * f, 0.0]) + rng.normal(0, 0.005, 3),
                         np.array([0.05 * f, 0.0, 0.8 * f]) + rng.normal(0, 0.02, 3)) for f in range(F)]
    mid = poses[F // 2]
    u = rng.uniform(60, width - 60, m)
    v = rng.uniform(30, height - 30, m)
    z = rng.uniform(6, 50, m)
    pc = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z, np.ones(m)])
    X = (mid @ pc)[:3]
    obs = [float(F), float(m)]
    for f in range(F):
        h = projection_matrix(K, poses[f]) @ np.vstack([X, np.ones(m)])
        uv = h[:2] / h[2]
        ok = (h[2] > 1) & (uv[0] > 0) & (uv[0] < width) & (uv[1] > 0) & (uv[1] < height) & (rng.random(m) < visible)
        idx = np.flatnonzero(ok)
        uvn = uv[:, idx] + rng.normal(0, noise_px, (2, len(idx)))
        obs.append(float(len(idx)))
        obs.extend(np.stack([uvn[1], uvn[0]], axis=1).reshape(-1))  # (row, col) interleaved
        obs.extend((idx + 1).astype(np.float64))
    state_true = np.concatenate([np.concatenate([twist_of(T) for T in poses]), X.T.reshape(-1)])
    state0 = state_true.copy()
    for f in range(F):
        Tp = poses[f] @ pose_matrix(rng.normal(0, rot_perturb, 3), rng.normal(0, trans_perturb, 3))
        state0[6 * f:6 * f + 6] = twist_of(Tp)
    state0[6 * F:] += rng.normal(0, lm_perturb, 3 * m)
    return state0, np.array(obs), state_true


def two_view_matches(n, outlier_ratio=0.2, noise_px=0.2, seed=0, K=K_KITTI):
    """Matched (row, col) points of a static scene seen from two poses, with gross outliers:
    returns p1, p2 (n, 2) and the boolean inlier labels."""
    rng = np.random.default_rng(seed)
    p1h, p2h, _, _, _ = two_view_problem(n, noise_px=noise_px, seed=seed + 1000, K=K, baseline=1.0)
    p1 = p1h[:2].T[:, ::-1].copy()
    p2 = p2h[:2].T[:, ::-1].copy()
    out = rng.random(n) < outlier_ratio
    p2[out] += rng.uniform(-60, 60, (out.sum(), 2))
    return p1, p2, ~out


class CorridorScene:
    """A piecewise-planar street canyon with procedural texture, ray-cast exactly: floor, two side
    walls, a ceiling far above and an end wall.  Camera convention: x right, y down, z forward
    (KITTI).  Gives frames of any size for a pose T_W_C and exact 3-D points for pixels, so
    synthetic sequences have known structure and motion (the reference's datasets are absent)."""

    def __init__(self, seed=5678, K=K_KITTI, width=KITTI_W, height=KITTI_H, length=400.0, half_width=7.0,
                 cam_height=1.65, roof=9.0, tile=1024, px_per_m=24.0):
        self.K, self.width, self.height = np.asarray(K, dtype=np.float64), width, height
        # planes as (normal n, offset d) with n.X = d; order: floor, left, right, roof, end
        self.planes = [(np.array([0.0, 1.0, 0.0]), cam_height), (np.array([1.0, 0.0, 0.0]), -half_width),
                       (np.array([1.0, 0.0, 0.0]), half_width), (np.array([0.0, 1.0, 0.0]), -roof),
                       (np.array([0.0, 0.0, 1.0]), length)]
        # in-plane axes used for the texture coordinates of each plane
        self.axes = [(0, 2), (2, 1), (2, 1), (0, 2), (0, 1)]
        self.tex = [textured_image(tile, tile, seed=seed + i, sigma=1.2, octaves=6) for i in range(5)]
        self.tile, self.ppm = tile, px_per_m
        v, u = np.mgrid[0:height, 0:width].astype(np.float64)
        self.rays_c = np.stack([(u - self.K[0, 2]) / self.K[0, 0], (v - self.K[1, 2]) / self.K[1, 1], np.ones_like(u)])

    def _cast(self, T_W_C, rays_c):
        R, t = T_W_C[:3, :3], T_W_C[:3, 3]
        d = np.tensordot(R, rays_c, axes=(1, 0))
        best = np.full(d.shape[1:], np.inf)
        which = np.zeros(d.shape[1:], dtype=np.int64)
        for i, (n, off) in enumerate(self.planes):
            den = np.tensordot(n, d, axes=(0, 0))
            with np.errstate(divide="ignore", invalid="ignore"):
                s = (off - n @ t) / den
            ok = (s > 1e-6) & (s < best)
            best = np.where(ok, s, best)
            which = np.where(ok, i, which)
        X = t.reshape(3, *([1] * (d.ndim - 1))) + d * best
        return X, which, best

    def render(self, T_W_C):
        """uint8 (height, width) view from T_W_C."""
        X, which, _ = self._cast(np.asarray(T_W_C, dtype=np.float64), self.rays_c)
        img = np.zeros(which.shape)
        for i, (a, b) in enumerate(self.axes):
            m = which == i
            if not m.any():
                continue
            tu = (X[a][m] * self.ppm) % self.tile
            tv = (X[b][m] * self.ppm) % self.tile
            iu, iv = np.floor(tu).astype(int), np.floor(tv).astype(int)
            fu, fv = tu - iu, tv - iv
            iu1, iv1 = (iu + 1) % self.tile, (iv + 1) % self.tile
            T = self.tex[i]
            img[m] = ((1 - fv) * ((1 - fu) * T[iv, iu] + fu * T[iv, iu1]) + fv * ((1 - fu) * T[iv1, iu] + fu * T[iv1, iu1]))
        return to_uint8(img)

    def backproject(self, T_W_C, xy):
        """World points (n, 3) seen at the 1-based pixel positions xy (n, 2) from T_W_C."""
        xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
        rays = np.stack([(xy[:, 0] - 1.0 - self.K[0, 2]) / self.K[0, 0], (xy[:, 1] - 1.0 - self.K[1, 2]) / self.K[1, 1],
                         np.ones(len(xy))])
        X, _, _ = self._cast(np.asarray(T_W_C, dtype=np.float64), rays)
        return X.T.copy()


def corridor_trajectory(n_frames, step=1.0, yaw_deg=0.5, period=100.0):
    """T_W_C per frame: `step` metres forward per frame with a sinusoidal yaw of +-yaw_deg."""
    poses = []
    for i in range(n_frames):
        yaw = np.deg2rad(yaw_deg) * np.sin(2 * np.pi * i / period)
        poses.append(pose_matrix(np.array([0.0, yaw, 0.0]), np.array([0.15 * np.sin(2 * np.pi * i / period), 0.0, step * i])))
    return poses


def corridor_sequence(n_frames, seed=5678, step=1.0, width=KITTI_W, height=KITTI_H, cache_dir=None):
    """Frames (list of uint8 arrays) and ground-truth poses of a corridor drive; rendered once and
    cached as .npy when cache_dir is given."""
    import os
    poses = corridor_trajectory(n_frames, step=step)
    path = None
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        path = os.path.join(cache_dir, "corridor_%d_%d_%dx%d_%g.npy" % (seed, n_frames, width, height, step))
        if os.path.exists(path):
            return list(np.load(path)), poses
    scene = CorridorScene(seed=seed, width=width, height=height)
    frames = [scene.render(T) for T in poses]
    if path:
        tmp = "%s.%d.tmp.npy" % (path, os.getpid())   # atomic: several ranks may render at once
        np.save(tmp, np.stack(frames))
        os.replace(tmp, path)
    return frames, poses
