"""Host-side mirror of utils/genKeypoints.m:1-33 for the detectors built natively
('SURF', 'Harris', 'MinEigen', 'FAST').  Detection runs in libvohip.so on the GPU; no CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib

DEFAULT_BORDER = 16     # px dropped at the image edge, the stand-in for extractFeatures' margin
DEFAULT_CAPACITY = 16384
DETECTORS = {"Harris": 0, "MinEigen": 1, "FAST": 2, "SURF": 3}   # rows of the dispatch table, genKeypoints.m:16-19
DEFAULT_MIN_QUALITY = {"Harris": 0.01, "MinEigen": 0.01, "FAST": 0.1, "SURF": 0.0}   # the toolbox defaults
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


def _detect_device_image(ctx, image, min_quality, filter_size, border, method, cap, min_contrast, metric_threshold):
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
    if method == "SURF":
        return SURFPoints(loc, metric[:total].cpu().numpy(), scratch["scale"][:total].cpu().numpy())
    return cornerPoints(loc, metric[:total].cpu().numpy())


def genKeypoints(image, method="SURF", options=None, ctx=None):
    """[features, valid_key_points] = genKeypoints(image, method, options)

    method 'SURF' (the default, as in genKeypoints.m:3-5) with options {'MetricThreshold', t}
    (main_eval.m:151,167), 'Harris' or 'MinEigen' with {'MinQuality', q, 'FilterSize', 5}, or 'FAST' with
    {'MinContrast', c, 'MinQuality', q} (main_eval.m:172-173).  Two extra options have no MATLAB
    counterpart: 'Border' (default 16) and 'Capacity'.  `features` is None: the
    continuous-operation loop never reads it (main_demo.m:377 discards it).  The other detectors
    of the reference's table (BRISK, ORB, SIFT) are closed-source toolbox functions
    that are not rebuilt here."""
    if method not in DETECTORS:
        raise NotImplementedError("genKeypoints: method '%s' is not built natively (%s are)"
                                  % (method, ", ".join(DETECTORS)))
    ctx = ctx or _lib.default_context()
    opts = _options(options)
    min_quality = float(opts.pop("MinQuality", DEFAULT_MIN_QUALITY[method])) if method != "SURF" else 0.0
    min_contrast = float(opts.pop("MinContrast", 0.2)) if method == "FAST" else 0.0
    metric_threshold = float(opts.pop("MetricThreshold", DEFAULT_METRIC_THRESHOLD)) if method == "SURF" else 0.0
    filter_size = int(opts.pop("FilterSize", 5)) if method in ("Harris", "MinEigen") else 5
    border = int(opts.pop("Border", DEFAULT_BORDER))
    cap = int(opts.pop("Capacity", DEFAULT_CAPACITY))
    if opts:
        raise ValueError("unknown option(s): %s" % ", ".join(opts))
    if hasattr(image, "data_ptr"):
        return None, _detect_device_image(ctx, image, min_quality, filter_size, border, method, cap, min_contrast,
                                          metric_threshold)
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
