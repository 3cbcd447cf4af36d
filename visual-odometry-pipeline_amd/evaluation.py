"""Host-side mirror of the reference's evaluation and result files:

  alignError                    utils/exe9_BA/alignError.m:1-17
  alignEstimateToGroundTruth    utils/exe9_BA/alignEstimateToGroundTruth.m:1-27
  quantitative_eval             utils/quantitative_eval.m:1-70 (the plotting at :29-50 is left out)
  experiment_name, save_results main_demo.m:453-463 and the save call at :473-476
  run_sequence                  the stand-alone run of main_demo.m:249-476 / one item of
                                main_eval.m:22-23: bootstrap, continuous operation, timing,
                                evaluation, result file
  main_eval                     the batch evaluation main_eval.m:20-23,483-494: the grid of independent
                                runs, sharded over one process per GPU, one result file per run

The alignment's non-linear least squares runs in libvohip.so on the GPU.  The result file is a
MAT v5 file with the key set the reference saves, which is what its evaluation notebook reads
(errs, fps_ct, frame_ct, toc_ct, S).
"""
import ctypes as C
import datetime
import time

import numpy as np

from . import _lib
from ._lib import colmajor, dptr
from .geometry import twist2HomogMatrix

ALIGN_MAX_ITER = 400      # lsqnonlin's defaults, which alignEstimateToGroundTruth.m:17-18 leaves in place:
ALIGN_FTOL = 1e-6         # MaxIterations 400, FunctionTolerance = StepTolerance = 1e-6
ALIGN_XTOL = 1e-6
DIST_CHECKPOINTS = np.array([10.0, 40.0, 90.0, 160.0, 250.0, 360.0]) * 0.2   # quantitative_eval.m:11-12


def alignError(x, pp_G_C, p_V_C, ctx=None):
    """error = alignError(x, pp_G_C, p_V_C): x = [twist; scale], points 3xN -> 3N-vector."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    T_G_V = twist2HomogMatrix(x[:6], ctx=ctx)
    p_G_C = x[6] * T_G_V[:3, :3] @ np.asarray(p_V_C, dtype=np.float64) + T_G_V[:3, 3:4]
    return (np.asarray(pp_G_C, dtype=np.float64) - p_G_C).reshape(-1, order="F")


def alignEstimateToGroundTruth(pp_G_C, p_V_C, return_x=False, ctx=None, max_iter=ALIGN_MAX_ITER, ftol=ALIGN_FTOL,
                               xtol=ALIGN_XTOL):
    """p_G_C = alignEstimateToGroundTruth(pp_G_C, p_V_C): the estimated trajectory p_V_C (3xN)
    under the similarity transform that best maps it onto the ground truth pp_G_C (3xN)."""
    ctx = ctx or _lib.default_context()
    g = colmajor(np.asarray(pp_G_C, dtype=np.float64).reshape(3, -1))
    e = colmajor(np.asarray(p_V_C, dtype=np.float64).reshape(3, -1))
    if g.shape != e.shape:
        raise ValueError("trajectories differ in length")
    n = g.shape[0]
    x = np.zeros(7)
    out = np.zeros((max(n, 1), 3))
    stats = np.zeros(3)
    ctx.check(_lib.load().vohip_align_sim3(ctx.handle, dptr(g), dptr(e), n, int(max_iter), float(ftol), float(xtol), dptr(x),
                                           dptr(out), dptr(stats)))
    p_G_C = out[:n].T.copy()
    if return_x:
        return p_G_C, x, dict(iterations=int(stats[0]), cost0=stats[1], cost=stats[2])
    return p_G_C


def quantitative_eval(ground_truth, S, bootstrap_frames, save_name=None, ctx=None):
    """errs = quantitative_eval(ground_truth, S, bootstrap_frames, save_name): ground_truth is the
    N-by-2 (x, z) table the scripts build, S.est_trans 3xT.  errs is 3-by-k, one column per distance
    checkpoint passed (k <= 6).  save_name only names the figure in the reference and is unused."""
    gt = np.asarray(ground_truth, dtype=np.float64)
    est = np.asarray(S.est_trans, dtype=np.float64)
    T = est.shape[1]
    b2 = int(bootstrap_frames[1])
    GT = gt[b2:b2 + T].T                                                    # :5
    if GT.shape[1] != T:
        raise ValueError("ground truth has %d rows after the bootstrap frame, the estimate %d" % (GT.shape[1], T))
    GT = np.vstack([GT[0], np.zeros(T), GT[1]])                             # :6
    dist = np.linalg.norm(GT[:, 1:] - GT[:, :-1], axis=0)                   # :9
    checkpoints = list(DIST_CHECKPOINTS)
    lengths = []                     # the prefixes the reference aligns, one per checkpoint passed
    current = 0.0
    nxt = checkpoints.pop(0)                                                # :16-17
    for i in range(1, T):                                                   # :20 (1-based i)
        current += dist[i - 1]
        if current > nxt:
            lengths.append(i)                                               # :23-24 align GT(:, 1:i), est(:, 1:i)
            if not checkpoints:                                             # :53-61
                break
            nxt = checkpoints.pop(0)
    if not lengths:
        return np.zeros((3, 0))
    # the alignments are independent of one another (each starts from the identity, :10-13): one launch for all
    ctx = ctx or _lib.default_context()
    g, e = colmajor(GT), colmajor(est)
    k = len(lengths)
    first_last = np.zeros((k, 6))
    ctx.check(_lib.load().vohip_align_sim3_prefixes(ctx.handle, dptr(g), dptr(e), T, k, (C.c_int * k)(*lengths), ALIGN_MAX_ITER,
                                                    ALIGN_FTOL, ALIGN_XTOL, None, dptr(first_last), None))
    errs = np.zeros((3, k))
    for b, i in enumerate(lengths):
        replacement_gt = GT[:, i - 1] - GT[:, 0]
        replacement_est = first_last[b, 3:] - first_last[b, :3]
        errs[:, b] = np.abs(replacement_est - replacement_gt)               # :25-27
    return errs


def experiment_name(dataset, hyper_paras, now=None):
    """main_demo.m:455-462: 'dataset@<name>_feat@<method>_time@HH_mm_ss[_NoRefined]'."""
    now = now or datetime.datetime.now()
    name = "dataset@%s_feat@%s_time@%s" % (dataset, hyper_paras.feature_extract, now.strftime("%H_%M_%S"))
    if not hyper_paras.is_refine_pose:
        name += "_NoRefined"
    return name


def _state_struct(S):
    col = lambda a: np.asarray(a, dtype=np.float64).reshape(-1, 1)          # noqa: E731
    return dict(P=S.P, X=S.X, C=S.C, F=S.F, T=S.T, est_trans=S.est_trans, est_rot=S.est_rot, num_X=col(S.num_X),
                num_C=col(S.num_C), num_new=col(S.num_new))


def save_results(save_name, errs, S, total_time, frame_init, frame_ct, toc_param, toc_bootstrap, toc_ct, fps_ct):
    """save([save_name '_res.mat'], 'errs', 'S', 'total_time', 'frame_init', 'frame_ct',
    'toc_param', 'toc_bootstrap', 'toc_ct', 'fps_ct') -- main_demo.m:473-476."""
    from scipy.io import savemat
    path = save_name + "_res.mat"
    mat = dict(S=_state_struct(S), total_time=float(total_time), frame_init=float(frame_init),
               frame_ct=float(frame_ct), toc_param=float(toc_param), toc_bootstrap=float(toc_bootstrap),
               toc_ct=float(toc_ct), fps_ct=float(fps_ct))
    if errs is not None:
        mat["errs"] = np.asarray(errs, dtype=np.float64)
    savemat(path, mat, oned_as="column")
    return path


def load_results(path):
    """The saved variables as a dict (S as a dict of arrays)."""
    from scipy.io import loadmat
    raw = loadmat(path, squeeze_me=False, struct_as_record=True)
    out = {k: v for k, v in raw.items() if not k.startswith("__")}
    if "S" in out:
        s = out["S"][0, 0]
        out["S"] = {name: s[name] for name in s.dtype.names}
    for k in ("total_time", "frame_init", "frame_ct", "toc_param", "toc_bootstrap", "toc_ct", "fps_ct"):
        if k in out:
            out[k] = float(np.asarray(out[k]).reshape(-1)[0])
    return out


def run_sequence(images, K, hyper_paras, ground_truth=None, bootstrap_frames=(0, 2), dataset="synthetic",
                 save_dir=None, rng=None, ctx=None, window_size=3, keyframe_d=2, now=None, device_loop=True,
                 prepare=True):
    """One full run as main_demo.m:249-476 does it: bootstrap on images[b1..b2], continuous
    operation on the remaining images, timing, evaluation against ground_truth (N-by-2 (x, z) table,
    optional) and, when save_dir is given, the '<experiment>_res.mat' file.
    Returns a dict with the saved variables plus 'S' (the final state) and 'path'.
    device_loop: the loop with its state resident on the GPU (one library call per frame, the following
    image of the stored sequence handed over ahead); False: the stage-by-stage loop, one call per
    reference function.
    prepare (device loop): the images of the loop go to the GPU in one go, with all their pyramids and corner
    lists from batched launches (VOPipelineDevice.prepare); the results do not depend on it."""
    from .bootstrap import bootstrap
    from .frame_pipeline import VOPipelineDevice
    from .pipeline import VOPipeline
    ctx = ctx or _lib.default_context()
    rng = rng or np.random.default_rng()
    t_param = time.perf_counter()
    b1, b2 = int(bootstrap_frames[0]), int(bootstrap_frames[1])
    K = np.asarray(K, dtype=np.float64)
    toc_param = time.perf_counter() - t_param
    t_boot = time.perf_counter()                                            # :162
    img0 = images[b1]
    img_seqs = list(images[b1 + 1:b2 + 1])                                  # :173-179
    p_W_landmarks, keypoints, T_init_WC = bootstrap(img0, img_seqs, K, hyper_paras, rng=rng, ctx=ctx)   # :252-264
    toc_bootstrap = time.perf_counter() - t_boot                            # :266
    t_ct = time.perf_counter()                                              # :270
    frame_range = range(b2 + 1, len(images))                                # :351
    if device_loop:
        pipe = VOPipelineDevice(K, img_seqs[-1], p_W_landmarks, keypoints, T_init_WC, hyper_paras=hyper_paras,
                                window_size=window_size, keyframe_d=keyframe_d, rng=rng, ctx=ctx)
    else:
        pipe = VOPipeline(K, img_seqs[-1], p_W_landmarks, keypoints, T_init_WC, hyper_paras=hyper_paras,
                          window_size=window_size, keyframe_d=keyframe_d, rng=rng, ctx=ctx)
    try:   # (a run that fails gives its pyramid slots and device state back all the same)
        if device_loop and prepare and len(frame_range) > 0:
            pipe.prepare([images[i] for i in frame_range])
        for i in frame_range:
            if device_loop:
                pipe.process_frame(images[i], next_image=images[i + 1] if i + 1 < len(images) else None)
            else:
                pipe.process_frame(images[i])
        toc_ct = time.perf_counter() - t_ct                                 # :446
        S = pipe.S
    finally:
        pipe.close()
    frame_init = len(img_seqs) + 1                                          # :447
    frame_ct = len(frame_range)                                             # :448
    fps_ct = frame_ct / toc_ct if toc_ct > 0 else 0.0                       # :449
    total_time = toc_param + toc_bootstrap + toc_ct                         # :451
    errs = None
    if ground_truth is not None:                                            # :465-471
        errs = quantitative_eval(ground_truth, S, (b1, b2), ctx=ctx)
    out = dict(errs=errs, S=S, total_time=total_time, frame_init=frame_init, frame_ct=frame_ct, toc_param=toc_param,
               toc_bootstrap=toc_bootstrap, toc_ct=toc_ct, fps_ct=fps_ct, path=None)
    if save_dir is not None:
        import os
        os.makedirs(save_dir, exist_ok=True)
        out["path"] = save_results(os.path.join(save_dir, experiment_name(dataset, hyper_paras, now)), errs, S,
                                   total_time, frame_init, frame_ct, toc_param, toc_bootstrap, toc_ct, fps_ct)
    return out


def run_group(runs, ctx=None, group=None, n_slots=None, max_frames=None):
    """Several full runs (each what run_sequence does: bootstrap, continuous operation, timing, evaluation,
    result file) whose loops advance together as ONE group of loops in lockstep (VOGroupDevice): every kernel
    of a frame is launched once for all of them.  The runs must share the image shape and camera matrix and
    differ at most in images, generator and corner detector (with its thresholds): the independent runs of
    main_eval.m:22-23.  A run's trajectory, counts, state and errs are those run_sequence
    gives it, bit for bit.

    runs: list of dicts with keys images, K, hyper_paras, rng and optionally ground_truth, bootstrap_frames,
    dataset, save_dir, now (the arguments of run_sequence).
    Timing: the runs bootstrap one after the other (toc_bootstrap is each run's own); the loops share their wall
    time, so a run's toc_ct is its share of the group's loop time by frames, and fps_ct = frame_ct / toc_ct
    is the rate at which the group as a whole processed frames.
    group: a VOGroupDevice to use (and keep open) instead of creating one with n_slots x max_frames.
    Returns one dict per run (the keys of run_sequence's result, plus 'error' for a run that stopped)."""
    from .bootstrap import bootstrap
    from .frame_pipeline import VOGroupDevice
    ctx = ctx or _lib.default_context()
    if not runs:
        return []
    boots = []
    for r in runs:
        t_param = time.perf_counter()
        b1, b2 = (int(v) for v in r.get("bootstrap_frames", (0, 2)))
        toc_param = time.perf_counter() - t_param
        t_boot = time.perf_counter()                                        # main_demo.m:162
        images = r["images"]
        img_seqs = list(images[b1 + 1:b2 + 1])
        p_W, kp, T0 = bootstrap(images[b1], img_seqs, np.asarray(r["K"], dtype=np.float64), r["hyper_paras"], rng=r["rng"],
                                ctx=ctx)                                    # :252-264
        boots.append(dict(b1=b1, b2=b2, p_W=p_W, kp=kp, T0=T0, toc_param=toc_param,
                          toc_bootstrap=time.perf_counter() - t_boot, frame_init=len(img_seqs) + 1))
    t_ct = time.perf_counter()                                              # :270
    own = group is None
    hp0 = runs[0]["hyper_paras"]
    if own:
        longest = max(len(r["images"]) - b["b2"] for r, b in zip(runs, boots))
        shape = tuple(np.asarray(runs[0]["images"][boots[0]["b2"]]).shape) if not hasattr(runs[0]["images"][0], "data_ptr") \
            else tuple(runs[0]["images"][0].shape)
        group = VOGroupDevice(runs[0]["K"], shape, n_slots or len(runs), max_frames or longest, hyper_paras=hp0, ctx=ctx)
    try:
        uploaded = {}   # (image list, first loop image) -> the sequence of the group that holds those images
        for s, (r, b) in enumerate(zip(runs, boots)):
            hp = r["hyper_paras"]
            fo = dict(zip(hp.feature_extract_options[::2], hp.feature_extract_options[1::2]))
            key = (id(r["images"]), b["b2"])
            # the loop's images: the last bootstrap image (the one the trackers are initialised on, :342-349),
            # then the frames of :351
            group.add_sequence(s, list(r["images"][b["b2"]:]), b["p_W"], b["kp"], b["T0"], rng=r["rng"],
                               feature_extract=hp.feature_extract, min_quality=fo.get("MinQuality"),
                               min_contrast=fo.get("MinContrast"), same_images_as=uploaded.get(key),
                               metric_threshold=fo.get("MetricThreshold"))
            uploaded.setdefault(key, s)
        group.run()
        loop_s = time.perf_counter() - t_ct                                 # :446
        states = [group.S(s) if group.seqs[s].error is None else None for s in range(len(runs))]
        errors = [group.seqs[s].error for s in range(len(runs))]
    finally:
        if own:
            group.close()
    total_frames = sum(len(r["images"]) - b["b2"] - 1 for r, b in zip(runs, boots))
    outs = []
    for r, b, S, err in zip(runs, boots, states, errors):
        frame_ct = len(r["images"]) - b["b2"] - 1                           # :448
        toc_ct = loop_s * frame_ct / max(total_frames, 1)
        out = dict(errs=None, S=S, total_time=b["toc_param"] + b["toc_bootstrap"] + toc_ct, frame_init=b["frame_init"],
                   frame_ct=frame_ct, toc_param=b["toc_param"], toc_bootstrap=b["toc_bootstrap"], toc_ct=toc_ct,
                   fps_ct=frame_ct / toc_ct if toc_ct > 0 else 0.0, path=None, error=err, group_loop_s=loop_s)
        if err is None:
            if r.get("ground_truth") is not None:                           # :465-471
                out["errs"] = quantitative_eval(r["ground_truth"], S, (b["b1"], b["b2"]), ctx=ctx)
            if r.get("save_dir") is not None:
                import os
                os.makedirs(r["save_dir"], exist_ok=True)
                name = experiment_name(r.get("dataset", "synthetic"), r["hyper_paras"], r.get("now"))
                out["path"] = save_results(os.path.join(r["save_dir"], name), out["errs"], S, out["total_time"],
                                           out["frame_init"], frame_ct, out["toc_param"], out["toc_bootstrap"], toc_ct,
                                           out["fps_ct"])
        outs.append(out)
    return outs


MAIN_EVAL_FEATS = ("SURF", "Harris", "BRISK", "FAST")         # main_eval.m:19
EVAL_FEATURE_OPTIONS = {"SURF": ["MetricThreshold", 200],     # main_eval.m:166-167
                        "BRISK": ["MinQuality", 2e-3],       # main_eval.m:170-171
                        "Harris": ["MinQuality", 2e-4],      # main_eval.m:168-169
                        "MinEigen": ["MinQuality", 1e-2],    # genKeypoints.m:18 (no entry in main_eval's switch)
                        "FAST": ["MinContrast", 0.11]}       # main_eval.m:172-173


def main_eval(sequences, feats=("Harris",), repetition=5, save_dir=None, seed=0, device=None, rank=None, world=None,
              ctx=None, now=None, concurrent=1, on_error="raise", group=0, balance=True, timing=None):
    """The batch evaluation of main_eval.m: for every sequence, every feature variant (main_eval.m:19,22,
    165-174) and every repetition (:20,23) one full, independent run (run_sequence: bootstrap, loop,
    timing, quantitative_eval, '<experiment>_res.mat' as :483-494 saves it).

    sequences: list of dicts with keys name, images, K, and optionally ground_truth (N x 2 (x, z) table),
    bootstrap_frames, hyper_paras (defaults: default_hyper_paras(name)), window_size, keyframe_d.
    The runs are the work items of batch.run_batch: dealt round-robin to one process per GPU
    (torch.distributed), no exchange while they run, one all_gather of the result records at the end, so
    every rank returns the full list of records (item order: sequence-major, then feature, then
    repetition).  rank / world without a process group: only that share is run (see batch.run_batch).

    MATLAB's repetitions differ only through the position of its never-seeded global RNG stream
    (main_eval.m:20-23); here run k draws from numpy.random.default_rng([seed, k]), so a run's result
    does not depend on how many processes share the grid.  The reference names a result file after the
    wall-clock second it is saved in; runs here take milliseconds, so run k is stamped `now` + k seconds
    to keep the names unique and parseable by the evaluation notebook.

    concurrent: runs a process keeps in flight on its GPU at once, one host thread and one context (own
    stream, own scratch) each; a single run leaves most of the GPU idle, so the grid finishes sooner, and a
    run's result does not depend on it.
    on_error: "raise" -- a run that fails (the localisation finds no pose, too few tracks: where the reference's
    script stops with an error, main_demo.m:389-391) ends the grid, as in the reference; "record" -- its record
    is marked failed (no trajectory) and the grid goes on.
    group: > 1 -- a rank advances up to that many of its runs together as a group of loops in lockstep
    (run_group: one launch sequence per frame for all of them; runs without bundle adjustment of one image
    shape and camera; the others run one at a time).  A run's record does not depend on it, apart
    from its timings (see run_group).
    balance: deal the runs longest first by frame count (batch.shard) instead of round-robin.
    timing: a dict that receives this rank's run and gather seconds."""
    import copy
    import datetime
    from . import batch
    from .pipeline import default_hyper_paras
    import threading
    ctx = ctx or _lib.default_context()
    now = now or datetime.datetime.now()
    items = [(si, feat, rep) for si in range(len(sequences)) for feat in feats for rep in range(repetition)]
    per_thread = threading.local()   # concurrent runs: a context per worker thread (a context serves one thread at a time)
    extra = []

    def context():
        if concurrent <= 1:
            return ctx
        if not hasattr(per_thread, "ctx"):
            per_thread.ctx = _lib.Context(ctx.device)
            extra.append(per_thread.ctx)
        return per_thread.ctx

    def runner(item_id, item):
        si, feat, rep = item
        seq = sequences[si]
        hp = copy.copy(seq.get("hyper_paras") or default_hyper_paras(seq["name"]))
        hp.feature_extract = feat                                           # main_eval.m:162-163
        hp.feature_extract_options = list(EVAL_FEATURE_OPTIONS[feat])       # :165-174
        t0 = time.perf_counter()
        try:
            out = run_sequence(seq["images"], seq["K"], hp, ground_truth=seq.get("ground_truth"),
                               bootstrap_frames=seq.get("bootstrap_frames", (0, 2)), dataset=seq["name"],
                               save_dir=save_dir,
                               rng=np.random.default_rng([seed, item_id]), ctx=context(),
                               window_size=seq.get("window_size", 3), keyframe_d=seq.get("keyframe_d", 2),
                               now=now + datetime.timedelta(seconds=item_id))
        except (RuntimeError, _lib.VohipError):
            if on_error != "record":
                raise
            return batch.failed_record(item_id, time.perf_counter() - t0)
        S = out["S"]
        return batch.pack_record(item_id, time.perf_counter() - t0, S.est_trans, S.est_rot, S.num_X, S.num_C,
                                 S.num_new, errs=out["errs"], timing=out)

    def groupable(item):
        si, feat, _ = item
        hp = sequences[si].get("hyper_paras") or default_hyper_paras(sequences[si]["name"])
        return not hp.is_BA

    def share_runner(mine):
        """This rank's share: groupable runs in groups of `group` (same shape and camera), the rest one by one."""
        out = {}
        buckets = {}
        for item_id, item in mine:
            if not groupable(item):
                out[item_id] = runner(item_id, item)
                continue
            seq = sequences[item[0]]
            key = (tuple(seq["images"][0].shape), np.asarray(seq["K"], dtype=np.float64).tobytes())
            buckets.setdefault(key, []).append((item_id, item))
        chunks = []
        for bucket in buckets.values():
            # (longest first, so that the runs of a group end together)
            bucket.sort(key=lambda p: (-len(sequences[p[1][0]]["images"]), p[0]))
            chunks += [bucket[c:c + int(group)] for c in range(0, len(bucket), int(group))]

        def do_chunk(chunk):
            runs = []
            for item_id, (si, feat, rep) in chunk:
                seq = sequences[si]
                hp = copy.copy(seq.get("hyper_paras") or default_hyper_paras(seq["name"]))
                hp.feature_extract = feat
                hp.feature_extract_options = list(EVAL_FEATURE_OPTIONS[feat])
                runs.append(dict(images=seq["images"], K=seq["K"], hyper_paras=hp, ground_truth=seq.get("ground_truth"),
                                 bootstrap_frames=seq.get("bootstrap_frames", (0, 2)), dataset=seq["name"],
                                 save_dir=save_dir, rng=np.random.default_rng([seed, item_id]),
                                 now=now + datetime.timedelta(seconds=item_id)))
            t0 = time.perf_counter()
            try:
                results = run_group(runs, ctx=context())
            except (RuntimeError, _lib.VohipError):   # (a bootstrap that fails: where the reference stops)
                if on_error != "record":
                    raise
                for item_id, item in chunk:   # one by one, so that the others of the group still get their record
                    out[item_id] = runner(item_id, item)
                return
            dt = time.perf_counter() - t0
            for (item_id, _), res in zip(chunk, results):
                if res["error"] is not None:
                    if on_error != "record":
                        raise res["error"]
                    out[item_id] = batch.failed_record(item_id, dt / len(chunk))
                    continue
                S = res["S"]
                out[item_id] = batch.pack_record(item_id, res["total_time"], S.est_trans, S.est_rot, S.num_X, S.num_C,
                                                 S.num_new, errs=res["errs"], timing=res)

        if concurrent > 1 and len(chunks) > 1:
            # groups in flight together, a host thread and a context each: one group's bootstraps, evaluation and
            # the idle stretches of its frames (localisation chain, tail) beside another group's trackers
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=int(concurrent)) as pool:
                for f in [pool.submit(do_chunk, c) for c in chunks]:
                    f.result()
        else:
            for c in chunks:
                do_chunk(c)
        return [out[item_id] for item_id, _ in mine]

    costs = [len(sequences[si]["images"]) for si, _, _ in items] if balance else None
    try:
        records = batch.run_batch(items, runner, device=device, rank=rank, world=world, concurrent=concurrent, costs=costs,
                                  share_runner=share_runner if group and group > 1 else None, timing=timing)
    finally:
        for c in extra:
            c.close()
    for (si, feat, rep), r in zip(items, records):
        if r is not None:
            r.update(sequence=sequences[si]["name"], feat=feat, repetition=rep)
    return records
