"""A second, structurally independent statement of the pyramidal Lucas-Kanade tracker, used only by
tests to pin the HIP kernel and the C oracle from outside their own arithmetic.

Plain numpy, everything in float64: double pyramids, double bilinear samples, straight np.sum over
the window (no lane banding, no fixed reduction tree, no float accumulators), vectorised over the
points instead of looping over them.  What it shares with the other two statements is only the
DEFINITION of the tracker that stands in for vision.PointTracker as the reference configures it
(main_demo.m:342-343: BlockSize [21 21], NumPyramidLevels 5, MaxIterations 20,
MaxBidirectionalError 6):
  * pyramid: [1 4 6 4 1]/16 along both axes, replicated border, even samples kept;
  * per level, coarse to fine: template and central-difference gradients of the previous image
    sampled bilinearly on the W x W window around the (sub-pixel) point, G = sum g g', Gauss-Newton
    steps d += G^-1 sum (T - J(p + d + w)) g, at most max_iter of them, stopping after a step shorter
    than 0.01 px or when a step cancels the previous one to within 0.01 px (half of it is then taken
    back); a level whose lambda_min(G) / W^2 < 1e-3 is skipped; the displacement doubles between levels;
  * valid iff level 0 is well conditioned in both directions, the forward result is inside the
    image and the forward-backward distance is <= max_bidir_error.
Images are indexed img[y, x]; points are (x, y), 0-based.
"""
import numpy as np

EPS_STEP = 0.01
MIN_EIG = 1e-3


def _down(img):
    """One pyramid reduction in float64."""
    k = np.array([1.0, 4.0, 6.0, 4.0, 1.0]) / 16.0
    p = np.pad(img, ((0, 0), (2, 2)), mode="edge")
    h = sum(k[j] * p[:, j:j + img.shape[1]] for j in range(5))[:, ::2]
    p = np.pad(h, ((2, 2), (0, 0)), mode="edge")
    return sum(k[j] * p[j:j + img.shape[0], :] for j in range(5))[::2, :]


def pyramid(img, levels):
    out = [np.asarray(img, dtype=np.float64)]
    for _ in range(1, levels):
        out.append(_down(out[-1]))
    return out


def _sample(img, x, y):
    """Bilinear samples of img at (x, y) (arrays of any common shape), replicated border."""
    H, W = img.shape
    x0 = np.floor(x)
    y0 = np.floor(y)
    fx = x - x0
    fy = y - y0
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    xa, xb = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    ya, yb = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    top = img[ya, xa] * (1.0 - fx) + img[ya, xb] * fx
    bot = img[yb, xa] * (1.0 - fx) + img[yb, xb] * fx
    return top * (1.0 - fy) + bot * fy


def _track_dir(P, Q, pts, W, max_iter):
    """pts n x 2 (x, y) in level-0 coordinates of pyramid P -> positions in Q and the level-0
    conditioning flag."""
    n = pts.shape[0]
    half = W // 2
    wy, wx = np.meshgrid(np.arange(-half, half + 1, dtype=np.float64), np.arange(-half, half + 1, dtype=np.float64),
                         indexing="ij")
    wx = wx[None]
    wy = wy[None]
    d = np.zeros((n, 2))
    ok0 = np.ones(n, dtype=bool)
    big = 1e6   # positions far outside the image cannot come back; keeps the integer casts bounded
    for l in range(len(P) - 1, -1, -1):
        c = pts / float(1 << l)
        cx = c[:, 0, None, None]
        cy = c[:, 1, None, None]
        T = _sample(P[l], cx + wx, cy + wy)
        gx = 0.5 * (_sample(P[l], cx + wx + 1.0, cy + wy) - _sample(P[l], cx + wx - 1.0, cy + wy))
        gy = 0.5 * (_sample(P[l], cx + wx, cy + wy + 1.0) - _sample(P[l], cx + wx, cy + wy - 1.0))
        gxx = (gx * gx).sum((1, 2))
        gxy = (gx * gy).sum((1, 2))
        gyy = (gy * gy).sum((1, 2))
        det = gxx * gyy - gxy * gxy
        lmin = 0.5 * ((gxx + gyy) - np.sqrt((gxx - gyy) ** 2 + 4.0 * gxy * gxy))
        good = (lmin / float(W * W) >= MIN_EIG) & (det > 0.0)
        if l == 0:
            ok0 = good
        active = good.copy()
        prev = np.zeros((n, 2))
        for it in range(max_iter):
            idx = np.flatnonzero(active)
            if idx.size == 0:
                break
            tx = np.clip(c[idx, 0] + d[idx, 0], -big, big)[:, None, None]
            ty = np.clip(c[idx, 1] + d[idx, 1], -big, big)[:, None, None]
            e = T[idx] - _sample(Q[l], tx + wx, ty + wy)
            bx = (e * gx[idx]).sum((1, 2))
            by = (e * gy[idx]).sum((1, 2))
            sx = (gyy[idx] * bx - gxy[idx] * by) / det[idx]
            sy = (gxx[idx] * by - gxy[idx] * bx) / det[idx]
            d[idx, 0] += sx
            d[idx, 1] += sy
            small = sx * sx + sy * sy < EPS_STEP * EPS_STEP
            osc = np.zeros_like(small)
            if it > 0:
                osc = ~small & ((sx + prev[idx, 0]) ** 2 + (sy + prev[idx, 1]) ** 2 < EPS_STEP * EPS_STEP)
                d[idx[osc], 0] -= 0.5 * sx[osc]
                d[idx[osc], 1] -= 0.5 * sy[osc]
            prev[idx, 0] = sx
            prev[idx, 1] = sy
            active[idx[small | osc]] = False
        if l > 0:
            d *= 2.0
    return pts + d, ok0


def track(prev, cur, pts, block=21, levels=5, max_iter=20, max_bidir_error=6.0):
    """prev, cur: 2-D uint8 (or float) images indexed [y, x]; pts n x 2 (x, y), 0-based.
    Returns (out n x 2, valid n bool, fb_err n)."""
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    P, Q = pyramid(prev, levels), pyramid(cur, levels)
    with np.errstate(all="ignore"):
        fwd, ok_f = _track_dir(P, Q, pts, block, max_iter)
        back, ok_b = _track_dir(Q, P, fwd, block, max_iter)
    H, W = np.asarray(prev).shape
    inside = (fwd[:, 0] >= 0.0) & (fwd[:, 0] <= W - 1.0) & (fwd[:, 1] >= 0.0) & (fwd[:, 1] <= H - 1.0)
    # (a point already invalid after the forward pass has no defined way back: infinite distance)
    err = np.where(ok_f & inside, np.linalg.norm(back - pts, axis=1), np.inf)
    with np.errstate(invalid="ignore"):
        valid = ok_f & ok_b & inside & (err <= max_bidir_error)
    return fwd, valid, err
