"""Host-side mirror of utils/genKeypoints.m:1-33 for the detectors built natively
('SURF', 'Harris', 'MinEigen', 'FAST', 'BRISK').  Detection runs in libvohip.so on the GPU; no CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib

DEFAULT_BORDER = 16     # px dropped at the image edge, the stand-in for extractFeatures' margin
DEFAULT_CAPACITY = 16384
DETECTORS = {"Harris": 0, "MinEigen": 1, "FAST": 2, "SURF": 3, "BRISK": 4}   # rows of the dispatch table, genKeypoints.m:16-20
DEFAULT_MIN_QUALITY = {"Harris": 0.01, "MinEigen": 0.01, "FAST": 0.1, "SURF": 0.0, "BRISK": 0.1}   # the toolbox defaults
DEFAULT_METRIC_THRESHOLD = 1000.0    # detectSURFFeatures' default


class cornerPoints:
    """The fields of MATLAB's cornerPoints that the pipeline reads: Location (N x 2 single,
    (x, y) 1-based), Metric (N single), Count."""

    def __init__(self, Location, Metric):
        self.Location = Location
        self.Metric = Metric

    @property
    def Count(self):
        return self.Location.shape[0]

    def __len__(self):
        return self.Count


class SURFPoints(cornerPoints):
    """The fields of MATLAB's SURFPoints that can be told from the detector alone: Location, Metric, Scale."""

    def __init__(self, Location, Metric, Scale):
        cornerPoints.__init__(self, Location, Metric)
        self.Scale = Scale


class BRISKPoints(cornerPoints):
    """Location, Metric and Scale (the scale-space level of the keypoint in units of the image's pixel)."""

    def __init__(self, Location, Metric, Scale):
        cornerPoints.__init__(self, Location, Metric)
        self.Scale = Scale


def _options(options):
    """MATLAB name-value cell {'MinQuality', 2e-4, ...} as list/tuple or dict."""
    if options is None:
        return {}
    if isinstance(options, dict):
        return dict(options)
    opts = list(options)
    if len(opts) % 2:
        raise ValueError("options must be name-value pairs")
    return {opts[i]: opts[i + 1] for i in range(0, len(opts), 2)}


def _detect_device_image(ctx, image, min_quality, filter_size, border, method, cap, min_contrast, metric_threshold,
                         num_octaves=4):
    """Detection on a uint8 image already resident on the GPU (an object with data_ptr() and
    shape, row-major): only the keypoint list crosses the bus."""
    import torch
    rows, cols = int(image.shape[0]), int(image.shape[1])
    scratch = ctx.__dict__.setdefault("_corner_scratch", {})
    if scratch.get("cap") != cap:
        scratch["cap"] = cap
        scratch["xy"] = torch.empty(2 * cap, dtype=torch.float32, device=image.device)
        scratch["metric"] = torch.empty(cap, dtype=torch.float32, device=image.device)
        scratch["count"] = torch.zeros(1, dtype=torch.int32, device=image.device)
        scratch["scale"] = torch.empty(cap, dtype=torch.float32, device=image.device)
    xy, metric, count = scratch["xy"], scratch["metric"], scratch["count"]
    if method == "SURF":
        ctx.check(_lib.load().vohip_detect_surf_dev(ctx.handle, image.data_ptr(), rows, cols, 0, metric_threshold, border,
                                                    xy.data_ptr(), metric.data_ptr(), scratch["scale"].data_ptr(), cap,
                                                    count.data_ptr()))
    elif method == "BRISK":
        ctx.check(_lib.load().vohip_detect_brisk_dev(ctx.handle, image.data_ptr(), rows, cols, 0, min_contrast, min_quality,
                                                     num_octaves, border, xy.data_ptr(), metric.data_ptr(),
                                                     scratch["scale"].data_ptr(), cap, count.data_ptr()))
    elif method == "FAST":
        ctx.check(_lib.load().vohip_detect_fast_dev(ctx.handle, image.data_ptr(), rows, cols, 0, min_contrast, min_quality,
                                                    border, xy.data_ptr(), metric.data_ptr(), cap, count.data_ptr()))
    else:
        ctx.check(_lib.load().vohip_detect_corners_dev(ctx.handle, image.data_ptr(), rows, cols, 0, min_quality,
                                                       filter_size, border, DETECTORS[method],
                                                       xy.data_ptr(), metric.data_ptr(), cap, count.data_ptr()))
    ctx.synchronize()
    total = int(count.item())
    if total > cap:
        raise _lib.VohipError(_lib.VOHIP_CAPACITY_EXCEEDED, "%d corners found, capacity %d" % (total, cap))
    loc = torch.stack([xy[:total], xy[cap:cap + total]], dim=1).cpu().numpy()
    if method in ("SURF", "BRISK"):
        return (SURFPoints if method == "SURF" else BRISKPoints)(loc, metric[:total].cpu().numpy(),
                                                                scratch["scale"][:total].cpu().numpy())
    return cornerPoints(loc, metric[:total].cpu().numpy())


def _host_image(image):
    """A uint8 image as (array in its own storage order, colmajor flag); device images are copied to the host."""
    if hasattr(image, "data_ptr"):
        image = image.cpu().numpy()
    img = np.asarray(image)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise TypeError("image must be a 2-D uint8 array")
    colmajor = bool(img.flags.f_contiguous and not img.flags.c_contiguous)
    if not (img.flags.c_contiguous or img.flags.f_contiguous):
        img = np.ascontiguousarray(img)
    return img, colmajor


def extractFeatures(image, points, Upright=False, ctx=None):
    """[features, validPoints] = extractFeatures(I, points) for SURFPoints (genKeypoints.m:32): N x 64 single SURF
    descriptors; validPoints are the points themselves with their Orientation filled in (boxes are clipped to the
    image, no point is dropped).  Other point types would get the toolbox's binary FREAK descriptors, which are
    not built: NotImplementedError."""
    if not isinstance(points, SURFPoints):
        raise NotImplementedError("extractFeatures is built for SURFPoints only (FREAK descriptors of corner points are not)")
    ctx = ctx or _lib.default_context()
    n = points.Count
    if n == 0:
        return np.zeros((0, 64), dtype=np.float32), points
    img, colmajor = _host_image(image)
    loc = np.ascontiguousarray(points.Location, dtype=np.float32)
    xy = np.ascontiguousarray(np.concatenate([loc[:, 0], loc[:, 1]]))
    scale = np.ascontiguousarray(points.Scale, dtype=np.float32)
    feats = np.zeros((n, 64), dtype=np.float32)
    ori = np.zeros(n, dtype=np.float32)
    ctx.check(_lib.load().vohip_extract_surf_features(
        ctx.handle, img.ctypes.data_as(_lib.c_uint8_p), img.shape[0], img.shape[1], int(colmajor),
        xy.ctypes.data_as(_lib.c_float_p), scale.ctypes.data_as(_lib.c_float_p), n, int(bool(Upright)),
        feats.ctypes.data_as(_lib.c_float_p), ori.ctypes.data_as(_lib.c_float_p)))
    valid = SURFPoints(points.Location, points.Metric, points.Scale)
    valid.Orientation = ori
    return feats, valid


def matchFeatures(features1, features2, MatchThreshold=1.0, MaxRatio=0.6, ctx=None):
    """indexPairs = matchFeatures(features1, features2) (matching_init.m:29) with the toolbox defaults for non-binary
    features: P x 2 index pairs, 0-BASED here (MATLAB's are 1-based), in the order of features1.  A second output,
    the pairs' distances, as matchFeatures' matchmetric."""
    ctx = ctx or _lib.default_context()
    f1 = np.ascontiguousarray(features1, dtype=np.float32)
    f2 = np.ascontiguousarray(features2, dtype=np.float32)
    if f1.ndim != 2 or f2.ndim != 2 or f1.shape[1] != f2.shape[1]:
        raise ValueError("feature sets must be N1 x D and N2 x D")
    if f1.shape[0] == 0 or f2.shape[0] == 0:
        return np.zeros((0, 2), dtype=np.int32), np.zeros(0, dtype=np.float32)
    pairs = np.zeros((f1.shape[0], 2), dtype=np.int32)
    metric = np.zeros(f1.shape[0], dtype=np.float32)
    m = C.c_int(0)
    ctx.check(_lib.load().vohip_match_features(
        ctx.handle, f1.ctypes.data_as(_lib.c_float_p), f1.shape[0], f2.ctypes.data_as(_lib.c_float_p), f2.shape[0],
        f1.shape[1], float(MatchThreshold), float(MaxRatio), pairs.ctypes.data_as(_lib.c_int_p),
        metric.ctypes.data_as(_lib.c_float_p), C.byref(m)))
    return pairs[:m.value].copy(), metric[:m.value].copy()


def genKeypoints(image, method="SURF", options=None, ctx=None, with_features=False):
    """[features, valid_key_points] = genKeypoints(image, method, options)

    method 'SURF' (the default, as in genKeypoints.m:3-5) with options {'MetricThreshold', t}
    (main_eval.m:151,167), 'Harris' or 'MinEigen' with {'MinQuality', q, 'FilterSize', 5}, or 'FAST' with
    {'MinContrast', c, 'MinQuality', q} (main_eval.m:172-173), or 'BRISK' with {'MinContrast', c, 'MinQuality', q,
    'NumOctaves', n} (main_eval.m:170-171).  Two extra options have no MATLAB
    counterpart: 'Border' (default 16) and 'Capacity'.  `features` is None (the continuous-operation loop never
    reads it, main_demo.m:377 discards it) unless with_features is set, as the 'Des_match' bootstrap branch does
    (matching_init.m:26-27): then the SURF descriptors of extractFeatures (SURF points only).  The other detectors
    of the reference's table (ORB, SIFT) are closed-source toolbox functions
    that are not rebuilt here."""
    if method not in DETECTORS:
        raise NotImplementedError("genKeypoints: method '%s' is not built natively (%s are)"
                                  % (method, ", ".join(DETECTORS)))
    ctx = ctx or _lib.default_context()
    opts = _options(options)
    min_quality = float(opts.pop("MinQuality", DEFAULT_MIN_QUALITY[method])) if method != "SURF" else 0.0
    min_contrast = float(opts.pop("MinContrast", 0.2)) if method in ("FAST", "BRISK") else 0.0
    num_octaves = int(opts.pop("NumOctaves", 4)) if method == "BRISK" else 4
    metric_threshold = float(opts.pop("MetricThreshold", DEFAULT_METRIC_THRESHOLD)) if method == "SURF" else 0.0
    filter_size = int(opts.pop("FilterSize", 5)) if method in ("Harris", "MinEigen") else 5
    border = int(opts.pop("Border", DEFAULT_BORDER))
    cap = int(opts.pop("Capacity", DEFAULT_CAPACITY))
    if opts:
        raise ValueError("unknown option(s): %s" % ", ".join(opts))
    if with_features:
        _, points = genKeypoints(image, method, options, ctx=ctx)
        return extractFeatures(image, points, ctx=ctx)                       # genKeypoints.m:32
    if hasattr(image, "data_ptr"):
        return None, _detect_device_image(ctx, image, min_quality, filter_size, border, method, cap, min_contrast,
                                          metric_threshold, num_octaves)
    img = np.asarray(image)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise TypeError("image must be a 2-D uint8 array")
    colmajor = bool(img.flags.f_contiguous and not img.flags.c_contiguous)
    if not (img.flags.c_contiguous or img.flags.f_contiguous):
        img = np.ascontiguousarray(img)
    xy = np.zeros(2 * cap, dtype=np.float32)
    metric = np.zeros(cap, dtype=np.float32)
    n = C.c_int(0)
    if method == "SURF":
        scale = np.zeros(cap, dtype=np.float32)
        ctx.check(_lib.load().vohip_detect_surf(
            ctx.handle, img.ctypes.data_as(_lib.c_uint8_p), img.shape[0], img.shape[1], int(colmajor), metric_threshold,
            border, xy.ctypes.data_as(_lib.c_float_p), metric.ctypes.data_as(_lib.c_float_p),
            scale.ctypes.data_as(_lib.c_float_p), cap, C.byref(n)))
        k = n.value
        return None, SURFPoints(np.stack([xy[:k], xy[k:2 * k]], axis=1).copy(), metric[:k].copy(), scale[:k].copy())
    if method == "BRISK":
        scale = np.zeros(cap, dtype=np.float32)
        ctx.check(_lib.load().vohip_detect_brisk(
            ctx.handle, img.ctypes.data_as(_lib.c_uint8_p), img.shape[0], img.shape[1], int(colmajor), min_contrast,
            min_quality, num_octaves, border, xy.ctypes.data_as(_lib.c_float_p), metric.ctypes.data_as(_lib.c_float_p),
            scale.ctypes.data_as(_lib.c_float_p), cap, C.byref(n)))
        k = n.value
        return None, BRISKPoints(np.stack([xy[:k], xy[k:2 * k]], axis=1).copy(), metric[:k].copy(), scale[:k].copy())
    if method == "FAST":
        ctx.check(_lib.load().vohip_detect_fast(
            ctx.handle, img.ctypes.data_as(_lib.c_uint8_p), img.shape[0], img.shape[1], int(colmajor), min_contrast,
            min_quality, border, xy.ctypes.data_as(_lib.c_float_p), metric.ctypes.data_as(_lib.c_float_p), cap,
            C.byref(n)))
    else:
        ctx.check(_lib.load().vohip_detect_corners(
            ctx.handle, img.ctypes.data_as(_lib.c_uint8_p), img.shape[0], img.shape[1], int(colmajor), min_quality,
            filter_size, border, DETECTORS[method], xy.ctypes.data_as(_lib.c_float_p),
            metric.ctypes.data_as(_lib.c_float_p), cap, C.byref(n)))
    k = n.value
    loc = np.stack([xy[:k], xy[k:2 * k]], axis=1).copy()
    return None, cornerPoints(loc, metric[:k].copy())
