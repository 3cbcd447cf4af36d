"""Host-side mirror of vision.PointTracker as the reference drives it
(main_demo.m:342-349, 380, 417-424; update_landmarks.m:10): constructor name-value pairs,
initialize(points, I), [points, validity] = tracker(I), release().  Tracking runs in
libvohip.so on the GPU (one wavefront per track); there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import dptr


class Pyramid:
    """A float image pyramid resident on the GPU (one of the context's 16 slots).  Build it once
    per frame and hand it to several trackers."""

    def __init__(self, image, levels=5, ctx=None):
        self.ctx = ctx or _lib.default_context()
        free = self.ctx.__dict__.setdefault("_free_pyr_slots", list(range(16)))
        if not free:
            raise RuntimeError("all 16 pyramid slots of the context are in use; release() some")
        self.slot = free.pop(0)
        if hasattr(image, "data_ptr"):
            # a uint8 image already resident on the GPU (e.g. a torch tensor), row-major
            shape = tuple(image.shape)
            self.shape, self.levels = shape, levels
            try:
                self.ctx.check(_lib.load().vohip_pyramid_build_dev(self.ctx.handle, self.slot, image.data_ptr(), shape[0],
                                                                   shape[1], 0, levels))
            except Exception:
                self.release()
                raise
            return
        img = np.asarray(image)
        if img.dtype != np.uint8 or img.ndim != 2:
            self.release()
            raise TypeError("image must be a 2-D uint8 array")
        # numpy C order = row-major; Fortran order = MATLAB's column-major, no copy either way
        colmajor = bool(img.flags.f_contiguous and not img.flags.c_contiguous)
        if not (img.flags.c_contiguous or img.flags.f_contiguous):
            img = np.ascontiguousarray(img)
        self.shape = img.shape
        self.levels = levels
        ptr = img.ctypes.data_as(_lib.c_uint8_p)
        try:
            self.ctx.check(_lib.load().vohip_pyramid_build(self.ctx.handle, self.slot, ptr, img.shape[0],
                                                           img.shape[1], int(colmajor), levels))
        except Exception:
            self.release()
            raise

    def level(self, l):
        """Level l as a float32 array indexed [slow axis, fast axis] (= [row, col] for C-order input)."""
        A, B = C.c_int(0), C.c_int(0)
        lib = _lib.load()
        self.ctx.check(lib.vohip_pyramid_level(self.ctx.handle, self.slot, l, None, C.byref(A), C.byref(B)))
        out = np.zeros((B.value, A.value), dtype=np.float32)
        self.ctx.check(lib.vohip_pyramid_level(self.ctx.handle, self.slot, l, out.ctypes.data_as(_lib.c_float_p),
                                               C.byref(A), C.byref(B)))
        return out

    def release(self):
        if self.slot is not None:
            self.ctx.__dict__.setdefault("_free_pyr_slots", []).append(self.slot)
            self.slot = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class PointTracker:
    """vision.PointTracker('BlockSize',[b b],'NumPyramidLevels',L,'MaxIterations',k,
    'MaxBidirectionalError',e).  Points are N-by-2 (x, y), 1-based, like MATLAB's."""

    def __init__(self, BlockSize=(31, 31), NumPyramidLevels=3, MaxIterations=30, MaxBidirectionalError=np.inf,
                 ctx=None):
        bs = tuple(int(b) for b in np.atleast_1d(BlockSize))
        if len(bs) != 2 or bs[0] != bs[1] or bs[0] not in (21, 31):
            raise ValueError("BlockSize must be [21 21] or [31 31]")
        self.BlockSize = bs
        self.NumPyramidLevels = int(NumPyramidLevels)
        self.MaxIterations = int(MaxIterations)
        self.MaxBidirectionalError = float(MaxBidirectionalError)
        self.ctx = ctx or _lib.default_context()
        self._prev = None
        self._own_prev = False
        self._points = None
        self.last_fb_error = None

    def _as_pyramid(self, I):
        if isinstance(I, Pyramid):
            if I.levels != self.NumPyramidLevels:
                raise ValueError("pyramid has %d levels, tracker wants %d" % (I.levels, self.NumPyramidLevels))
            return I, False
        return Pyramid(I, self.NumPyramidLevels, ctx=self.ctx), True

    def initialize(self, points, I):
        """initialize(tracker, points, I)"""
        self.release()
        self._prev, self._own_prev = self._as_pyramid(I)
        self._points = np.array(points, dtype=np.float64).reshape(-1, 2)
        self._valid = np.ones(self._points.shape[0], dtype=bool)

    def __call__(self, I):
        """[points, validity] = tracker(I).  A point that was lost stays lost in later steps
        (the tracker only follows valid points), until initialize() is called again."""
        if self._prev is None:
            raise RuntimeError("PointTracker: call initialize() before stepping")
        cur, own = self._as_pyramid(I)
        n = self._points.shape[0]
        pts = np.ascontiguousarray(self._points.T)  # column-major N x 2
        out = np.zeros((2, max(n, 1)))
        valid = np.zeros(max(n, 1), dtype=np.uint8)
        fb = np.zeros(max(n, 1))
        e = self.MaxBidirectionalError
        self.ctx.check(_lib.load().vohip_klt_track(
            self.ctx.handle, self._prev.slot, cur.slot, dptr(pts), n, self.BlockSize[0], self.MaxIterations,
            e if np.isfinite(e) else 1e300, dptr(out), valid.ctypes.data_as(_lib.c_uint8_p), dptr(fb)))
        if self._own_prev:
            self._prev.release()
        self._prev, self._own_prev = cur, own
        points = out[:, :n].T.copy()
        validity = valid[:n].astype(bool) & self._valid
        self._valid = validity
        self._points = points
        self.last_fb_error = fb[:n]
        return points, validity

    step = __call__

    def release(self):
        """release(tracker)"""
        if self._prev is not None and self._own_prev:
            self._prev.release()
        self._prev = None
        self._points = None
