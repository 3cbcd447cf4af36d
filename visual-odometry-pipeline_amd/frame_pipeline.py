"""The continuous-operation loop (main_demo.m:377-427, main_BA.m:261-305) with its state resident on
the GPU: a thin driver over the vohip_frame_* entry points of libvohip.so.  Same constructor and
methods as pipeline.VOPipeline; the state S is downloaded only when asked for.

With hyper_paras.is_BA the window bookkeeping of VO_bundle_adjust.m stays on the host, as in the
reference, inside the library (vohip_frame_bundle_adjust: S.P and S.X come down in one round trip, the
window struct B lives in a vohip_window, runBA and the normal-frame re-refinements run on the GPU, and
when a window was optimised the landmarks and the keyframe's pose go back up); this driver only applies
the poses the call hands back to its trajectory.

Per frame the host does one library call (one synchronisation) and draws the random numbers the two
RANSAC stages consume, from the same two generators as VOPipeline: `rng` gives 3 uniforms per P3P
hypothesis (it ends up advanced by exactly n_iter triples, as ransacLocalization would leave it),
`rng_map` 8 per fundamental-matrix trial.
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np

from . import _lib
from ._lib import VOHIP_NO_MODEL, VOHIP_SAMPLES_EXHAUSTED, colmajor, dptr
from .keypoints import DEFAULT_MIN_QUALITY, DETECTORS
from .pipeline import default_hyper_paras
from .ransac import MAX_ITERATIONS


class VOPipelineDevice:
    def __init__(self, K, first_image, p_W_landmarks, keypoints_xy, T_init_WC, hyper_paras=None, rng=None, ctx=None,
                 tracker_options=None, rng_map=None, cap_landmarks=8192, cap_candidates=8192, cap_corners=16384,
                 window_size=3, keyframe_d=2):
        self.ctx = ctx or _lib.default_context()
        self.K = np.asarray(K, dtype=np.float64)
        self.hp = hyper_paras or default_hyper_paras()
        if self.hp.feature_extract not in DETECTORS:
            raise NotImplementedError("detector '%s' is not built" % self.hp.feature_extract)
        self.rng = rng or np.random.default_rng()
        self.rng_map = rng_map or self.rng.spawn(1)[0]
        # PCG64 steps backwards by a modular advance (one step per double drawn)
        self._rewind = type(self.rng.bit_generator).__name__ in ("PCG64", "PCG64DXSM")
        lib = _lib.load()
        prm = _lib.FrameParams()
        lib.vohip_frame_default_params(C.byref(prm))
        opts = dict(BlockSize=[21, 21], NumPyramidLevels=5, MaxIterations=20, MaxBidirectionalError=6)  # main_demo.m:342-349
        opts.update(tracker_options or {})
        prm.klt_block = int(np.atleast_1d(opts["BlockSize"])[0])
        prm.klt_levels = int(opts["NumPyramidLevels"])
        prm.klt_max_iter = int(opts["MaxIterations"])
        e = float(opts["MaxBidirectionalError"])
        prm.klt_max_bidir_error = e if np.isfinite(e) else 1e300
        fo = dict(zip(self.hp.feature_extract_options[::2], self.hp.feature_extract_options[1::2]))
        prm.detector = DETECTORS[self.hp.feature_extract]
        prm.min_quality = float(fo.get("MinQuality", DEFAULT_MIN_QUALITY[self.hp.feature_extract]))
        prm.min_contrast = float(fo.get("MinContrast", 0.2))
        prm.metric_threshold = float(fo.get("MetricThreshold", 1000.0))
        prm.brisk_octaves = int(fo.get("NumOctaves", 4))
        prm.filter_size = int(fo.get("FilterSize", 5))
        prm.r_discard_redundant = float(self.hp.r_discard_redundant)
        prm.angle_threshold_deg = float(self.hp.angle_threshold)
        prm.min_depth = float(self.hp.min_depth)
        prm.max_depth = float(self.hp.max_depth)
        prm.refine_pose = int(bool(self.hp.is_refine_pose))
        prm.is_BA = int(bool(self.hp.is_BA))
        prm.cap_landmarks, prm.cap_candidates, prm.cap_corners = int(cap_landmarks), int(cap_candidates), int(cap_corners)
        free = self.ctx.__dict__.setdefault("_free_pyr_slots", list(range(16)))
        if len(free) < 3:
            raise RuntimeError("the context has fewer than 3 free pyramid slots")
        self._slots = [free.pop(), free.pop(), free.pop()]
        prm.slot_a, prm.slot_b, prm.slot_c = self._slots
        self.params = prm
        shape = tuple(first_image.shape)
        self._colmajor = 0
        if not hasattr(first_image, "data_ptr"):
            first_image = np.asarray(first_image)
            self._colmajor = int(first_image.flags.f_contiguous and not first_image.flags.c_contiguous)
        self._shape = shape
        h = C.c_void_p()
        try:
            self.ctx.check(lib.vohip_frame_create(self.ctx.handle, dptr(colmajor(self.K)), shape[0], shape[1], self._colmajor,
                                                  C.byref(prm), C.byref(h)))
        except Exception:
            free.extend(self._slots)   # (a loop that was not made holds no pyramid slots)
            self._slots = []
            raise
        self._h = h
        _lib._live_frames.add(self)
        X = np.ascontiguousarray(np.asarray(p_W_landmarks, dtype=np.float64).reshape(-1, 3))
        P_rc = np.ascontiguousarray(np.asarray(keypoints_xy, dtype=np.float64).reshape(-1, 2)[:, ::-1])
        T = np.asarray(T_init_WC, dtype=np.float64)
        ptr, on_dev, keep = self._image_ptr(first_image)
        self.ctx.check(lib.vohip_frame_reset(self._h, ptr, on_dev, dptr(X), None, dptr(P_rc), X.shape[0],
                                             dptr(colmajor(T))))
        self._keep_images = (keep, None)
        self._next_image = None
        self._u8_ahead = None
        self._pending = None
        self.est_trans = [T[:3, 3].copy()]
        self.est_rot = [T[:3, :3].reshape(-1, order="F").copy()]
        self.num_X, self.num_C, self.num_new = [X.shape[0]], [0], [0]
        self.frames = 0
        self.first_hypotheses = 104   # supplied with the frame; the adaptive bound asks for more when it needs them
        self._window = None
        self.ba_runs = 0
        if self.hp.is_BA:   # main_demo.m:288-303: the window struct B, kept by the library
            wp = _lib.WindowParams()
            lib.vohip_window_default_params(C.byref(wp))
            wp.window_size, wp.keyframe_d = int(window_size), int(keyframe_d)
            wp.min_depth, wp.max_depth = float(self.hp.min_depth), float(self.hp.max_depth)
            wh = C.c_void_p()
            self.ctx.check(lib.vohip_window_create(self.ctx.handle, dptr(colmajor(self.K)), C.byref(wp), C.byref(wh)))
            self._window = wh
            self.ctx.check(lib.vohip_window_reset(wh, dptr(X), dptr(P_rc), X.shape[0], dptr(colmajor(T))))
            self._ba_cap = wp.window_size * (wp.keyframe_d + 1)
            self._ba_offsets = (C.c_int * self._ba_cap)()
            self._ba_poses = np.zeros((self._ba_cap, 16))
        self.last_pose = None
        self.last_counts = None

    def _image_ptr(self, image):
        # an image that prepare() had to copy (wrong memory order, not an ndarray): the library knows the copy
        hit = getattr(self, "_prepared_as", {}).get(id(image))
        if hit is not None and hit[0] is image:
            self._prepared_hits += 1
            return C.c_void_p(hit[1].ctypes.data), 0, hit[1]
        if hasattr(image, "data_ptr"):                      # uint8 image already on the GPU (torch tensor)
            if tuple(image.shape) != self._shape:
                raise ValueError("image shape changed")
            return C.c_void_p(image.data_ptr()), 1, image
        img = np.asarray(image)
        if img.dtype != np.uint8 or img.shape != self._shape:
            raise TypeError("image must be a uint8 array of shape %s" % (self._shape,))
        cm = int(img.flags.f_contiguous and not img.flags.c_contiguous)
        if cm != self._colmajor:
            img = np.asfortranarray(img) if self._colmajor else np.ascontiguousarray(img)
        return C.c_void_p(img.ctypes.data), 0, img

    def prepare(self, images):
        """For a stored sequence: all of its images go to the GPU once, and their pyramids and corner lists are
        made by batched launches (vohip_frame_prepare).  Frames later passed to process_frame / begin_frame as
        the same objects then cost no transfer, no pyramid and no detection of their own; results do not change.
        Pass an empty list to release the prepared data."""
        self._warn_unused_prepare()
        self._prepared_as = {}
        ptrs, keep, on_dev, copies = [], [], None, {}
        for im in images:
            ptr, dev, k = self._image_ptr(im)
            if on_dev is None:
                on_dev = dev
            if dev != on_dev:
                raise ValueError("host and device images cannot be mixed in one prepared sequence")
            ptrs.append(ptr.value)
            keep.append(k)
            if k is not im:   # converted: later calls with `im` must find this very copy, not make another
                copies[id(im)] = (im, k)
        arr = (C.c_void_p * max(len(ptrs), 1))(*ptrs)
        self.ctx.check(_lib.load().vohip_frame_prepare(self._h, arr, len(ptrs), int(on_dev or 0)))
        # (the pointers are the library's keys: the arrays must stay alive, in place and unchanged)
        self._prepared = keep
        self._prepared_as = copies
        self._prepared_hits = 0
        self._prepared_ptrs = set(ptrs)

    def _warn_unused_prepare(self):
        """A prepared sequence none of whose images was ever passed to the loop holds GPU memory for nothing: the
        images were probably copied or reallocated between prepare() and the loop (the library recognises a
        prepared image by its address)."""
        done = len(self.num_X) - 1
        if getattr(self, "_prepared", None) and getattr(self, "_prepared_hits", 0) == 0 and done > getattr(self, "_prepared_at", 0):
            import warnings
            warnings.warn("VOPipelineDevice.prepare(): none of the %d prepared images was passed to the loop since "
                          "(pass the same objects that were prepared)" % len(self._prepared), RuntimeWarning, stacklevel=3)
        self._prepared_at = done

    def process_frame(self, image, next_image=None):
        """One frame of the loop (main_demo.m:377-427).  next_image: the frame after this one, when the
        caller already holds it (a stored sequence): its transfer, pyramid and corner detection then run
        beside this frame's localisation; the results do not change, and the next call must pass that image."""
        self.begin_frame(image, next_image)
        return self.finish_frame()

    def begin_frame(self, image, next_image=None):
        """First half of process_frame: enqueues the frame on the GPU and returns at once (several loops
        that share a context or a GPU can have their frames in flight together)."""
        lib = _lib.load()
        if self._pending is not None:
            raise RuntimeError("a frame is already in flight: finish_frame() first")
        if self._next_image is not None:
            if image is not self._next_image[0]:
                # the library has that image's pyramid, corners and trackers under way already
                raise ValueError("the previous call prepared its next_image ahead: pass that same object as image")
            ptr, on_dev, keep = self._next_image[1]          # handed over ahead by the previous call
        else:
            ptr, on_dev, keep = self._image_ptr(image)
        nptr, non_dev, nkeep = self._image_ptr(next_image) if next_image is not None else (None, 0, None)
        self._next_image = (next_image, (nptr, non_dev, nkeep)) if next_image is not None else None
        if getattr(ptr, "value", None) in getattr(self, "_prepared_ptrs", ()):
            self._prepared_hits += 1
        have_map = self.num_C[-1] > 0                                                       # :406
        u8 = None
        if have_map:                                                                        # update_landmarks.m:13
            u8 = self._u8_ahead if self._u8_ahead is not None else self.rng_map.random((self.params.fm_trials, 8))
        self._u8_ahead = None
        # (a generator that can step backwards is rewound after the frame; any other is saved and restored)
        state = None if self._rewind else self.rng.bit_generator.state
        have = int(self.first_hypotheses)
        u3 = self.rng.random((have, 3))                                                     # :387-388
        self.ctx.check(lib.vohip_frame_begin(self._h, ptr, on_dev, nptr, non_dev, dptr(u3), have,
                                             dptr(u8) if have_map else None))
        # while the GPU works on this frame: the trial draws of the next one (its generator is independent
        # of the localisation's; they are given back if the next frame turns out to have no candidates)
        state_map = self.rng_map.bit_generator.state
        u8_ahead = self.rng_map.random((self.params.fm_trials, 8))
        self._pending = (state, have, have_map, state_map, u8_ahead, keep, nkeep)

    def finish_frame(self):
        """Second half of process_frame: waits for the frame and returns T_W_C."""
        lib = _lib.load()
        if self._pending is None:
            raise RuntimeError("no frame is in flight: begin_frame() first")
        state, have, have_map, state_map, u8_ahead, keep, nkeep = self._pending
        self._pending = None
        pose, counts = _lib.FramePose(), _lib.FrameCounts()
        st = lib.vohip_frame_finish(self._h, C.byref(pose), C.byref(counts))
        drawn = have
        while st == VOHIP_SAMPLES_EXHAUSTED and have < MAX_ITERATIONS:
            have = min(max(8 * have, int(np.ceil(pose.num_iterations)) + 8), MAX_ITERATIONS)
            if state is None:
                self.rng.bit_generator.advance(-3 * drawn)
            else:
                self.rng.bit_generator.state = state
            u3 = self.rng.random((have, 3))         # the same leading draws, then more
            drawn = have
            st = lib.vohip_frame_retry(self._h, dptr(u3), have, C.byref(pose), C.byref(counts))
        self._keep_images = (keep, nkeep)   # a host image handed over ahead is read until the next call returns
        # the generator as n_iter executed iterations leave it (ransacLocalization.m:46-47 draws one triple each)
        if state is None:
            if drawn != pose.n_iter:
                self.rng.bit_generator.advance(-3 * (drawn - max(pose.n_iter, 0)))
        else:
            self.rng.bit_generator.state = state
            if pose.n_iter > 0:
                self.rng.random((pose.n_iter, 3))
        if st != 0:
            self.rng_map.bit_generator.state = state_map
        if st == VOHIP_NO_MODEL:
            raise RuntimeError("ransacLocalization found no pose (the reference fails here too, main_demo.m:389-391)")
        self.ctx.check(st)
        T_W_C = np.array(pose.T_W_C).reshape(4, 4).T.copy()
        self.est_trans.append(T_W_C[:3, 3].copy())                                          # :401-403
        self.est_rot.append(T_W_C[:3, :3].reshape(-1, order="F").copy())
        if have_map:
            self.num_new.append(counts.num_new)                                             # update_landmarks.m:124
        self.num_X.append(counts.n_landmarks)                                               # :426-427
        self.num_C.append(counts.n_candidates)
        if counts.n_candidates > 0:
            self._u8_ahead = u8_ahead
        else:
            self.rng_map.bit_generator.state = state_map
        self.frames += 1
        self.last_pose, self.last_counts = pose, counts
        if self.hp.is_BA:                                                                   # :409-411
            self._bundle_adjust(T_W_C)
        return T_W_C

    def _bundle_adjust(self, T_W_C):
        """VO_bundle_adjust.m:1-98 for the frame just finished; the poses it rewrites (:53-57, :80-84) come back
        as (column offset from the end of the trajectory, T_W_C) pairs."""
        opt, npose = C.c_int(0), C.c_int(0)
        self.ctx.check(_lib.load().vohip_frame_bundle_adjust(self._h, self._window, dptr(colmajor(T_W_C)), C.byref(opt),
                                                             C.byref(npose), self._ba_offsets, dptr(self._ba_poses),
                                                             self._ba_cap))
        if not opt.value:
            return
        self.ba_runs += 1
        last = len(self.est_trans) - 1
        for k in range(npose.value):
            T = self._ba_poses[k].reshape(4, 4).T
            col = last + self._ba_offsets[k]
            self.est_trans[col] = T[:3, 3].copy()
            self.est_rot[col] = T[:3, :3].reshape(-1, order="F").copy()

    @property
    def B(self):
        """The window struct of main_demo.m:288-303 as the library holds it: num_key, m, tau (6 x num_key),
        landmarks (4 x m)."""
        if self._window is None:
            return None
        lib = _lib.load()
        nk, m = C.c_int(0), C.c_int(0)
        self.ctx.check(lib.vohip_window_state(self._window, C.byref(nk), C.byref(m), None, None))
        tau, lm = np.zeros((max(nk.value, 1), 6)), np.zeros((max(m.value, 1), 4))
        self.ctx.check(lib.vohip_window_state(self._window, None, None, dptr(tau), dptr(lm)))
        return SimpleNamespace(num_key=nk.value, m=m.value, tau=tau[:nk.value].T.copy(), landmarks=lm[:m.value].T.copy())

    def trajectory(self):
        """est_trans (3 x T) and est_rot (9 x T), as saved by main_demo.m:473-476."""
        return np.array(self.est_trans).T.copy(), np.array(self.est_rot).T.copy()

    @property
    def S(self):
        """The state struct of main_demo.m:272-286, downloaded from the GPU."""
        lib = _lib.load()
        nL, nC = C.c_int(0), C.c_int(0)
        self.ctx.check(lib.vohip_frame_sizes(self._h, C.byref(nL), C.byref(nC)))
        nL, nC = nL.value, nC.value
        P, X, ids = np.zeros((max(nL, 1), 2)), np.zeros((max(nL, 1), 3)), np.zeros(max(nL, 1))
        Cc, F, T = np.zeros((max(nC, 1), 2)), np.zeros((max(nC, 1), 2)), np.zeros((max(nC, 1), 12))
        self.ctx.check(lib.vohip_frame_state(self._h, dptr(P), dptr(X), dptr(ids), dptr(Cc), dptr(F), dptr(T)))
        et, er = self.trajectory()
        return SimpleNamespace(P=P[:nL].T.copy(), X=np.vstack([X[:nL].T, ids[:nL]]), C=Cc[:nC].T.copy(), F=F[:nC].T.copy(),
                               T=T[:nC].T.copy(), est_trans=et, est_rot=er, num_X=list(self.num_X), num_C=list(self.num_C),
                               num_new=list(self.num_new))

    def close(self):
        if getattr(self, "_h", None) is not None and getattr(self, "num_X", None):
            self._warn_unused_prepare()
        if getattr(self, "_window", None) is not None:
            _lib.load().vohip_window_destroy(self._window)
            self._window = None
        if getattr(self, "_h", None) is not None:
            _lib.load().vohip_frame_destroy(self._h)
            self._h = None
            self.ctx.__dict__.setdefault("_free_pyr_slots", []).extend(self._slots)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VOGroupDevice:
    """Several independent loops of one image shape, camera matrix and parameter set, advanced in lockstep:
    one library call (vohip_group_begin / _finish) per frame for ALL of them, every kernel launched once with
    the sequence as a grid dimension.  This is how the independent runs of main_eval.m:22-23 (feature variant x
    repetition, each the loop of main_demo.m:377-427 over a stored sequence) share one GPU: a single loop
    leaves most of it idle and is bound by launch latencies and the host's enqueue work, which a group shares.

    Every sequence keeps its own two generators and its own state; its counts, ids and poses are those of a
    VOPipelineDevice run over the same images with the same generators, bit for bit.

    add_sequence(images, ...): images[0] is the image the loop starts from (main_demo.m:342-344: the last
    bootstrap image), images[1:] are its frames; all of them go to the GPU at once, with their pyramids and
    corner lists from batched launches.  run() processes every frame of every sequence; step() one frame."""

    def __init__(self, K, image_shape, n_sequences, max_frames, hyper_paras=None, ctx=None, tracker_options=None,
                 cap_landmarks=8192, cap_candidates=8192, cap_corners=16384, colmajor_images=False):
        self.ctx = ctx or _lib.default_context()
        self.K = np.asarray(K, dtype=np.float64)
        self.hp = hyper_paras or default_hyper_paras()
        if self.hp.is_BA:
            raise NotImplementedError("bundle adjustment is not available in a group of loops")
        if self.hp.feature_extract not in DETECTORS:
            raise NotImplementedError("detector '%s' is not built" % self.hp.feature_extract)
        lib = _lib.load()
        prm = _lib.FrameParams()
        lib.vohip_frame_default_params(C.byref(prm))
        opts = dict(BlockSize=[21, 21], NumPyramidLevels=5, MaxIterations=20, MaxBidirectionalError=6)  # main_demo.m:342-349
        opts.update(tracker_options or {})
        prm.klt_block = int(np.atleast_1d(opts["BlockSize"])[0])
        prm.klt_levels = int(opts["NumPyramidLevels"])
        prm.klt_max_iter = int(opts["MaxIterations"])
        e = float(opts["MaxBidirectionalError"])
        prm.klt_max_bidir_error = e if np.isfinite(e) else 1e300
        fo = dict(zip(self.hp.feature_extract_options[::2], self.hp.feature_extract_options[1::2]))
        prm.detector = DETECTORS[self.hp.feature_extract]
        prm.min_quality = float(fo.get("MinQuality", DEFAULT_MIN_QUALITY[self.hp.feature_extract]))
        prm.min_contrast = float(fo.get("MinContrast", 0.2))
        prm.metric_threshold = float(fo.get("MetricThreshold", 1000.0))
        prm.brisk_octaves = int(fo.get("NumOctaves", 4))
        prm.filter_size = int(fo.get("FilterSize", 5))
        prm.r_discard_redundant = float(self.hp.r_discard_redundant)
        prm.angle_threshold_deg = float(self.hp.angle_threshold)
        prm.min_depth = float(self.hp.min_depth)
        prm.max_depth = float(self.hp.max_depth)
        prm.refine_pose = int(bool(self.hp.is_refine_pose))
        prm.is_BA = 0
        prm.cap_landmarks, prm.cap_candidates, prm.cap_corners = int(cap_landmarks), int(cap_candidates), int(cap_corners)
        free = self.ctx.__dict__.setdefault("_free_pyr_slots", list(range(16)))
        if len(free) < 3:
            raise RuntimeError("the context has fewer than 3 free pyramid slots")
        self._slots = [free.pop(), free.pop(), free.pop()]
        prm.slot_a, prm.slot_b, prm.slot_c = self._slots
        self.params = prm
        self._shape = tuple(image_shape)
        self._colmajor = int(bool(colmajor_images))
        self.n_sequences = int(n_sequences)
        self.max_frames = int(max_frames)
        h = C.c_void_p()
        try:
            self.ctx.check(lib.vohip_group_create(self.ctx.handle, dptr(colmajor(self.K)), self._shape[0], self._shape[1],
                                                  self._colmajor, C.byref(prm), self.n_sequences, self.max_frames,
                                                  C.byref(h)))
        except Exception:
            free.extend(self._slots)   # (a group that was not made holds no pyramid slots)
            self._slots = []
            raise
        self._h = h
        _lib._live_frames.add(self)
        self.first_hypotheses = 104
        self.seqs = [None] * self.n_sequences
        self.frames = 0                 # frames processed by the run under way
        self.host_enqueue_us = 0.0      # summed over the frames of the run
        self.host_wait_us = 0.0

    def add_sequence(self, seq, images, p_W_landmarks, keypoints_xy, T_init_WC, rng=None, rng_map=None,
                     feature_extract=None, min_quality=None, min_contrast=None, same_images_as=None,
                     metric_threshold=None):
        """Sequence `seq` of the group: its stored images, the state its loop starts from (as VOPipelineDevice's
        constructor takes it) and its generators.  feature_extract / min_quality / min_contrast / metric_threshold:
        this sequence's detector when it differs from the group's (the feature variants of main_eval.m:165-174).
        same_images_as: a sequence of this run of the group that was added with these very images (repetitions and
        feature variants run over one image sequence): they are copied on the GPU instead of uploaded again."""
        if self.frames:
            self.frames = 0   # (the library starts a new run with the first reset after a frame)
            self.host_enqueue_us = self.host_wait_us = 0.0
            self.seqs = [None] * self.n_sequences
        lib = _lib.load()
        if len(images) < 2 or len(images) > self.max_frames:
            raise ValueError("a sequence holds 2..max_frames images")
        det = -1 if feature_extract is None else DETECTORS[feature_extract]
        mq = float(min_quality) if min_quality is not None else (
            DEFAULT_MIN_QUALITY[feature_extract] if feature_extract is not None else 0.0)
        mc = float(min_contrast) if min_contrast is not None else 0.2     # detectFASTFeatures' default
        if feature_extract == "SURF":   # (the library takes the detector's one other threshold in the same argument)
            mc = float(metric_threshold) if metric_threshold is not None else 1000.0   # detectSURFFeatures' default
        ptrs, keep, on_dev = [], [], None
        other = self.seqs[same_images_as] if same_images_as is not None else None
        if other is not None:
            if other.n_frames != len(images):
                raise ValueError("the other sequence holds a different number of images")
            same = (det, mq, mc) == other.detector
            self.ctx.check(lib.vohip_group_prepare_like(self._h, int(seq), int(same_images_as), -2 if same else det, mq, mc))
            images = []
        for im in images:
            if hasattr(im, "data_ptr"):
                if tuple(im.shape) != self._shape:
                    raise ValueError("image shape differs from the group's")
                ptr, dev, k = im.data_ptr(), 1, im
            else:
                img = np.asarray(im)
                if img.dtype != np.uint8 or img.shape != self._shape:
                    raise TypeError("images must be uint8 arrays of shape %s" % (self._shape,))
                img = np.asfortranarray(img) if self._colmajor else np.ascontiguousarray(img)
                ptr, dev, k = img.ctypes.data, 0, img
            if on_dev is None:
                on_dev = dev
            if dev != on_dev:
                raise ValueError("host and device images cannot be mixed in one sequence")
            ptrs.append(ptr)
            keep.append(k)
        if other is None:
            arr = (C.c_void_p * len(ptrs))(*ptrs)
            self.ctx.check(lib.vohip_group_prepare(self._h, int(seq), arr, len(ptrs), int(on_dev), det, mq, mc))
        n_images = other.n_frames if other is not None else len(ptrs)
        X = np.ascontiguousarray(np.asarray(p_W_landmarks, dtype=np.float64).reshape(-1, 3))
        P_rc = np.ascontiguousarray(np.asarray(keypoints_xy, dtype=np.float64).reshape(-1, 2)[:, ::-1])
        T = np.asarray(T_init_WC, dtype=np.float64)
        self.ctx.check(lib.vohip_group_reset(self._h, int(seq), dptr(X), None, dptr(P_rc), X.shape[0], dptr(colmajor(T))))
        rng = rng or np.random.default_rng()
        rng_map = rng_map or rng.spawn(1)[0]
        self.seqs[seq] = SimpleNamespace(
            n_frames=n_images, detector=(det, mq, mc), rng=rng, rng_map=rng_map,
            rewind=type(rng.bit_generator).__name__ in ("PCG64", "PCG64DXSM"),
            est_trans=[T[:3, 3].copy()], est_rot=[T[:3, :3].reshape(-1, order="F").copy()],
            num_X=[X.shape[0]], num_C=[0], num_new=[0], error=None, last_pose=None, last_counts=None)

    def active(self):
        """Indices of the sequences that take part in the next frame."""
        k = self.frames + 1
        return [i for i, q in enumerate(self.seqs) if q is not None and q.error is None and k < q.n_frames]

    def step(self):
        """One frame of every active sequence.  Returns {sequence: T_W_C}; a sequence whose frame fails (where
        the reference's script stops with an error, main_demo.m:389-391) records the exception in
        seqs[i].error and takes no further part."""
        self.begin_step()
        return self.finish_step()

    def begin_step(self):
        """The first half of step(): draws the frame's uniforms and enqueues it.  A host that drives several
        groups (each on a context of its own) begins one group's frame while another's is on the GPU."""
        lib = _lib.load()
        act = self.active()
        if not act:
            raise RuntimeError("no sequence of the group has a frame left")
        S, T = self.n_sequences, self.params.fm_trials
        have = int(self.first_hypotheses)
        have_map = self.seqs[act[0]].num_C[-1] > 0                                          # main_demo.m:406
        u3 = np.zeros((S, have, 3))
        u8 = np.zeros((S, T, 8)) if have_map else None
        states = {}
        for i in act:
            q = self.seqs[i]
            if have_map:                                                                    # update_landmarks.m:13
                u8[i] = q.rng_map.random((T, 8))
            states[i] = None if q.rewind else q.rng.bit_generator.state
            u3[i] = q.rng.random((have, 3))                                                 # :387-388
        self.ctx.check(lib.vohip_group_begin(self._h, dptr(u3), have, dptr(u8) if have_map else None))
        self._begun = (act, have, have_map, states)

    def finish_step(self):
        """The second half of step(): waits for the frame begin_step() enqueued and books its results."""
        lib = _lib.load()
        if getattr(self, "_begun", None) is None:
            raise RuntimeError("finish_step() without begin_step()")
        (act, have, have_map, states), self._begun = self._begun, None
        S = self.n_sequences
        poses = (_lib.FramePose * S)()
        counts = (_lib.FrameCounts * S)()
        status = (C.c_int * S)()
        self.ctx.check(lib.vohip_group_finish(self._h, poses, counts, status))
        self.frames += 1
        out = {}
        for i in act:
            q = self.seqs[i]
            st, pose, cnt = status[i], poses[i], counts[i]
            drawn, state = have, states[i]
            while st == VOHIP_SAMPLES_EXHAUSTED and drawn < MAX_ITERATIONS:
                more = min(max(8 * drawn, int(np.ceil(pose.num_iterations)) + 8), MAX_ITERATIONS)
                if state is None:
                    q.rng.bit_generator.advance(-3 * drawn)
                else:
                    q.rng.bit_generator.state = state
                u = q.rng.random((more, 3))         # the same leading draws, then more
                drawn = more
                pose, cnt = _lib.FramePose(), _lib.FrameCounts()
                st = lib.vohip_group_retry(self._h, i, dptr(u), more, C.byref(pose), C.byref(cnt))
            # the generator as n_iter executed iterations leave it (ransacLocalization.m:46-47)
            if state is None:
                if drawn != pose.n_iter:
                    q.rng.bit_generator.advance(-3 * (drawn - max(pose.n_iter, 0)))
            else:
                q.rng.bit_generator.state = state
                if pose.n_iter > 0:
                    q.rng.random((pose.n_iter, 3))
            q.last_pose, q.last_counts = pose, cnt
            if st != 0:
                msg = "sequence %d stopped in frame %d (%s)" % (i, self.frames, _lib.STATUS_NAMES.get(st, st))
                q.error = RuntimeError(msg) if st == VOHIP_NO_MODEL else _lib.VohipError(st, msg)
                continue
            T_W_C = np.array(pose.T_W_C).reshape(4, 4).T.copy()
            q.est_trans.append(T_W_C[:3, 3].copy())                                         # :401-403
            q.est_rot.append(T_W_C[:3, :3].reshape(-1, order="F").copy())
            if have_map:
                q.num_new.append(cnt.num_new)                                               # update_landmarks.m:124
            q.num_X.append(cnt.n_landmarks)                                                 # :426-427
            q.num_C.append(cnt.n_candidates)
            out[i] = T_W_C
        i0 = act[0]
        self.host_enqueue_us += counts[i0].host_enqueue_us
        self.host_wait_us += counts[i0].host_wait_us
        return out

    def run(self):
        """Every frame of every sequence.  Returns the number of (sequence, frame) pairs processed."""
        done = 0
        while self.active():
            done += len(self.step())
        return done

    def trajectory(self, seq):
        q = self.seqs[seq]
        return np.array(q.est_trans).T.copy(), np.array(q.est_rot).T.copy()

    def S(self, seq):
        """The state struct of main_demo.m:272-286 of sequence `seq`, downloaded from the GPU."""
        lib = _lib.load()
        q = self.seqs[seq]
        nL, nC = C.c_int(0), C.c_int(0)
        self.ctx.check(lib.vohip_group_sizes(self._h, int(seq), C.byref(nL), C.byref(nC), None))
        nL, nC = nL.value, nC.value
        P, X, ids = np.zeros((max(nL, 1), 2)), np.zeros((max(nL, 1), 3)), np.zeros(max(nL, 1))
        Cc, F, T = np.zeros((max(nC, 1), 2)), np.zeros((max(nC, 1), 2)), np.zeros((max(nC, 1), 12))
        self.ctx.check(lib.vohip_group_state(self._h, int(seq), dptr(P), dptr(X), dptr(ids), dptr(Cc), dptr(F), dptr(T)))
        et, er = self.trajectory(seq)
        return SimpleNamespace(P=P[:nL].T.copy(), X=np.vstack([X[:nL].T, ids[:nL]]), C=Cc[:nC].T.copy(), F=F[:nC].T.copy(),
                               T=T[:nC].T.copy(), est_trans=et, est_rot=er, num_X=list(q.num_X), num_C=list(q.num_C),
                               num_new=list(q.num_new))

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.load().vohip_group_destroy(self._h)
            self._h = None
            self.ctx.__dict__.setdefault("_free_pyr_slots", []).extend(self._slots)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
