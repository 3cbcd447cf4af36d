#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (N = 1): the per-frame continuous-operation loop on a rendered KITTI-shape sequence
(1241 x 376, ~2k tracked keypoints, Harris MinQuality 2e-4, 21x21 / 5-level KLT, adaptive P3P
RANSAC, pose refinement, F-matrix filter, triangulation, candidate filter; no BA) -- one step = one
frame of one sequence, frames resident in HBM before the timed region; the loop state stays on the GPU
(VOPipelineDevice over the vohip_frame_* entry points).  N > 1: the reference's batch mode
(main_eval.m:22-23) shards independent runs, so every rank runs its own work item of the run grid --
its own rendered sequence and random seed, dealt by batch.shard -- with no exchange while it runs, and
the result records are gathered with one all_gather (RCCL) inside the timed region (the collective's
first use happens in the set-up run before it; its duration is reported).  At N > 1 the line also
carries the batch mode proper: every rank advancing four runs of the grid as a group of loops.

Also measured and reported in the same JSON line (N = 1): the same loop with its images in pinned host
memory (upload included), with bundle adjustment (main_BA.m's window), in the parking configuration, at
stress sizes; 4, 8 and 16 independent sequences advanced as a group of loops in lockstep (one launch
sequence per frame for all of them: what the batch evaluation runs on a GPU) and four sequences on four host
threads; the batch evaluation itself (main_eval); the roofline entry of the tracker kernel (`roofline`:
the loop's own launches, event-timed in an untimed replay); the RANSAC stress configuration (4096 P3P
hypotheses x 8192 correspondences per step) as hypotheses/s with the roofline of its scoring kernel
(`roofline_ransac`).  The CPU oracle is used only for the cpu_baseline leg.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "VO frames/sec (KITTI 1241x376, 2k kpts) + RANSAC hypotheses/sec per GPU"
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector fp64: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X vector fp32
LDS_PEAK_TBS = 78.6            # 256 CU x 128 B/clk x 2.4 GHz
MATLAB_KITTI_HARRIS_FPS = 3.817  # the reference's published figure (Ryzen 7 5800H, MATLAB, real KITTI 05)
CACHE = os.environ.get("VOHIP_CACHE", "/tmp/vohip_cache")
MATLAB_PARKING_BA_FPS = 1.681    # ... parking with bundle adjustment (window 5, every 2nd frame), 5-run mean
MATLAB_PARKING_FPS = 4.559       # ... parking without bundle adjustment, 5-run mean
# Workload sizes.  KITTI configuration: about 2 000 tracked landmarks and about 2 000 candidates per frame are
# held with a 10 px thinning cell for the first frame's corners and a 5.5 px exclusion radius for new
# candidates (the reference's 10 px leaves ~1 100 + ~950 on the rendered corridor); Harris MinQuality stays 2e-4.
KITTI_THIN_CELL = 10.0
KITTI_EXCLUSION_RADIUS = 5.5
# Stress configuration: every corner the rendered frames hold (about 8 250 local maxima at MinQuality 1e-6)
# with a 2 px exclusion radius: about 6 400 landmarks + 5 400 candidates tracked per frame.
STRESS_MIN_QUALITY = 1e-6
STRESS_EXCLUSION_RADIUS = 2.0
STRESS_CAPACITY = 16384


def thin_by_grid(xy, metric, cell=None):
    """Strongest corner per cell: brings the first frame's detections to about 2k points."""
    cell = cell or KITTI_THIN_CELL
    key = (np.floor(xy[:, 1] / cell).astype(np.int64) << 20) + np.floor(xy[:, 0] / cell).astype(np.int64)
    order = np.lexsort((-metric, key))
    first = np.ones(len(order), dtype=bool)
    first[1:] = key[order][1:] != key[order][:-1]
    return np.sort(order[first])


def make_sequence(vo, n_frames, item=0):
    """Work item `item` of the batch grid: its own rendered corridor (texture seed 5678 + item)."""
    frames, poses = vo.synth.corridor_sequence(n_frames, seed=5678 + item, cache_dir=CACHE)
    scene = vo.synth.CorridorScene(seed=5678 + item)
    return scene, frames, poses


def init_points(vo, ctx, scene, frame0, pose0, thin=True, min_quality=2e-4, limit=None, feature=None):
    _, kp = vo.genKeypoints(frame0, *(feature or ("Harris", ["MinQuality", min_quality])), ctx=ctx)
    xy = kp.Location.astype(np.float64)
    if thin:
        xy = xy[thin_by_grid(xy, kp.Metric)]
    elif limit and len(xy) > limit:   # the strongest `limit` corners, in detection order
        xy = xy[np.sort(np.argsort(-kp.Metric, kind="stable")[:limit])]
    return scene.backproject(pose0, xy), xy


def bench_frames(vo, torch, ctx, scene, frames, poses, steps, warmup, seed, barrier, gather, resident=True, ahead=True,
                 stress_hypotheses=0, bundle_adjust=False, feature=None):
    """ahead: every call also hands over the following frame of the stored sequence, whose pyramid and corners
    are then prepared beside the localisation of the current one (same results; the sequence holds one frame
    more than is processed).
    stress_hypotheses > 0: the stress configuration -- every corner of the first frame starts as a landmark
    (about 6k, no thinning) and that many P3P hypotheses are solved and scored in every frame (the adaptive
    rule still picks the winner among the iterations it would have executed).
    bundle_adjust: main_BA.m's configuration (is_BA, window of 5 keyframes, every 2nd frame a keyframe).
    feature: (method, options) of genKeypoints.m:16-19 when not Harris with main_eval.m:169's MinQuality."""
    dev = torch.device("cuda", torch.cuda.current_device())
    if resident:
        d_frames = [torch.from_numpy(f).to(dev) for f in frames[:warmup + steps + 2]]
    else:   # frames stay in host memory; every step uploads its 466 KB image
        # pinned, so that the transfers are asynchronous
        keep = [torch.from_numpy(f).pin_memory() for f in frames[:warmup + steps + 2]]
        for t in keep:   # the first transfer out of a fresh pinned buffer maps its pages (~100 us): not the loop's cost
            t.to(dev)
        torch.cuda.synchronize()
        d_frames = [t.numpy() for t in keep]
    hp = vo.default_hyper_paras("kitti", is_BA=bundle_adjust)
    hp.r_discard_redundant = KITTI_EXCLUSION_RADIUS
    if stress_hypotheses:   # every corner of the frames and a narrow exclusion radius: about 12k points tracked per frame
        hp.feature_extract_options = ["MinQuality", STRESS_MIN_QUALITY]
        hp.r_discard_redundant = STRESS_EXCLUSION_RADIUS
    if feature:
        hp.feature_extract, hp.feature_extract_options = feature[0], list(feature[1])
    X0, xy0 = init_points(vo, ctx, scene, frames[0], poses[0], thin=not stress_hypotheses,
                          min_quality=STRESS_MIN_QUALITY if stress_hypotheses else 2e-4, limit=STRESS_CAPACITY,
                          feature=feature)
    cap = STRESS_CAPACITY if stress_hypotheses else 8192
    pipe = vo.VOPipelineDevice(vo.synth.K_KITTI, d_frames[0], X0, xy0, poses[0], hyper_paras=hp,
                               rng=np.random.default_rng(seed), ctx=ctx, cap_landmarks=cap, cap_candidates=cap,
                               window_size=5, keyframe_d=1)
    if stress_hypotheses:
        pipe.first_hypotheses = stress_hypotheses
    for i in range(1, warmup + 1):
        pipe.process_frame(d_frames[i], next_image=d_frames[i + 1] if ahead else None)
    barrier()
    t0 = time.perf_counter()
    enq = wait = 0.0
    for i in range(warmup + 1, warmup + steps + 1):
        pipe.process_frame(d_frames[i], next_image=d_frames[i + 1] if ahead else None)
        enq += pipe.last_counts.host_enqueue_us
        wait += pipe.last_counts.host_wait_us
    records = gather(pipe, time.perf_counter() - t0)
    barrier()
    elapsed = time.perf_counter() - t0
    last = warmup + steps
    T = pipe.est_trans[-1]
    info = dict(n_landmarks=int(np.mean(pipe.num_X[-steps:])), n_candidates=int(np.mean(pipe.num_C[-steps:])),
                final_position_error_m=float(np.linalg.norm(T - poses[last][:3, 3])), records=len(records),
                host_enqueue_us_per_frame=enq / steps, host_wait_us_per_frame=wait / steps, ba_runs=pipe.ba_runs)
    pipe.close()
    return elapsed, info


def bench_frames_concurrent(vo, torch, scene, frames, poses, steps, warmup, n_seq):
    """Batch evaluation on one GPU (main_eval.m:22-23 runs independent sequences): n_seq sequences at the same
    time, one host thread, context and stream set per sequence; returns the aggregate frames/s.  A single
    sequence leaves most of the GPU idle (every stage is latency-bound), so independent sequences overlap."""
    import threading
    dev = torch.device("cuda", torch.cuda.current_device())
    d_frames = [torch.from_numpy(f).to(dev) for f in frames[:warmup + steps + 2]]
    seconds = [0.0] * n_seq
    gate = threading.Barrier(n_seq)
    errors = []

    def worker(idx):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                ctx = vo.Context(dev.index, stream=stream.cuda_stream)
                X0, xy0 = init_points(vo, ctx, scene, frames[0], poses[0])
                hp = vo.default_hyper_paras("kitti", is_BA=False)
                hp.r_discard_redundant = KITTI_EXCLUSION_RADIUS
                pipe = vo.VOPipelineDevice(vo.synth.K_KITTI, d_frames[0], X0, xy0, poses[0], hyper_paras=hp,
                                           rng=np.random.default_rng(1000 + idx), ctx=ctx)
                for i in range(1, warmup + 1):
                    pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
                gate.wait()
                t0 = time.perf_counter()
                for i in range(warmup + 1, warmup + steps + 1):
                    pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
                seconds[idx] = time.perf_counter() - t0
                gate.wait()
                pipe.close()
                ctx.close()
        except Exception as e:   # a failed worker must not leave the others at the barrier
            errors.append(e)
            gate.abort()

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_seq)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return n_seq * steps / max(seconds)


def kitti_hyper_paras(vo):
    hp = vo.default_hyper_paras("kitti", is_BA=False)
    hp.r_discard_redundant = KITTI_EXCLUSION_RADIUS
    return hp


def bench_group(vo, torch, ctx, scene, frames, poses, steps, warmup, n_seq, seed=1000, measure_tracker=False, groups=1):
    """A group of n_seq loops in lockstep (VOGroupDevice: the independent runs of main_eval.m:22-23 sharing every
    launch of a frame) on the headline workload.  Sequence s starts s frames into the rendered corridor (its own
    images, start set and random draws).  One step = one frame of every sequence; returns the aggregate frames/s.
    groups > 1: the sequences as that many groups, each on a context of its own, driven in turn by this one host
    thread (a group's frame is begun while the other groups' frames are on the GPU, so one group's localisation
    chain and tail, which leave the GPU idle, run beside another's trackers).
    measure_tracker: the same run once more with the tracker launches bracketed by timing events and once
    more with the tracker's work counters on (neither while timing) -> the roofline entry of those launches."""
    import ctypes as C
    lib = vo._lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    n = warmup + steps + 1
    if len(frames) < n + n_seq - 1:
        raise ValueError("the rendered sequence is too short for %d staggered sequences" % n_seq)
    if n_seq % groups or (measure_tracker and groups != 1):
        raise ValueError("bad split of the sequences")
    d_frames = [torch.from_numpy(f).to(dev) for f in frames[:n + n_seq - 1]]
    starts = [init_points(vo, ctx, scene, frames[s], poses[s]) for s in range(n_seq)]
    per = n_seq // groups
    ctxs = [ctx] + [vo.Context(torch.cuda.current_device()) for _ in range(groups - 1)]
    gs = [vo.VOGroupDevice(vo.synth.K_KITTI, tuple(frames[0].shape), per, n, hyper_paras=kitti_hyper_paras(vo), ctx=c)
          for c in ctxs]
    g = gs[0]

    def run(timed):
        for s in range(n_seq):
            gs[s // per].add_sequence(s % per, d_frames[s:s + n], starts[s][0], starts[s][1], poses[s],
                                      rng=np.random.default_rng(seed + s))
        for _ in range(warmup):
            for q in gs:
                q.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, w0 = sum(q.host_enqueue_us for q in gs), sum(q.host_wait_us for q in gs)
        pairs = 0
        for q in gs:
            q.begin_step()
        for k in range(steps):
            for q in gs:
                pairs += len(q.finish_step())
                if k + 1 < steps:
                    q.begin_step()
        torch.cuda.synchronize()
        return (pairs, time.perf_counter() - t0, sum(q.host_enqueue_us for q in gs) - e0,
                sum(q.host_wait_us for q in gs) - w0)

    run(False)   # (code objects, staging buffers, the group's first use of its memory)
    pairs, dt, enq, wait = run(True)
    seqs = [q for grp in gs for q in grp.seqs]
    out = {"sequences": n_seq, "groups": groups, "frames_per_s": pairs / dt, "ms_per_step": 1e3 * dt / steps,
           "host_enqueue_us_per_step": enq / steps, "host_wait_us_per_step": wait / steps,
           "tracked_landmarks_per_sequence": int(np.mean([q.num_X[-1] for q in seqs])),
           "candidates_per_sequence": int(np.mean([q.num_C[-1] for q in seqs])),
           "final_position_error_m": float(max(np.linalg.norm(q.est_trans[-1] - poses[s + warmup + steps][:3, 3])
                                               for s, q in enumerate(seqs)))}
    if measure_tracker:
        tot, longest, launches = C.c_double(0), C.c_double(0), C.c_int(0)
        ctx.set_option("klt_time", 1)
        run(False)
        ctx.check(lib.vohip_klt_time(ctx.handle, C.byref(tot), C.byref(longest), C.byref(launches), 1))
        ctx.set_option("klt_time", 0)
        tr, it = C.c_longlong(0), C.c_longlong(0)
        ctx.set_option("klt_stats", 1)
        ctx.check(lib.vohip_klt_stats(ctx.handle, C.byref(tr), C.byref(it), 1))
        run(False)
        ctx.check(lib.vohip_klt_stats(ctx.handle, C.byref(tr), C.byref(it), 1))
        ctx.set_option("klt_stats", 0)
        out["tracker"] = tracker_roofline(tot.value, longest.value, launches.value, tr.value, it.value,
                                          frames[0].shape, frames_of_run=(warmup + steps) * n_seq, group=n_seq > 1)
    for q in gs:
        q.close()
    for c in ctxs[1:]:
        c.close()
    return out


def tracker_roofline(total_ms, longest_ms, launches, tracks, iterations, shape, frames_of_run, group=False):
    """Roofline entry of the tracker launches of a loop (or group of loops): the launches' own durations (event pairs
    on their streams) against the work their tracks did (the kernel's counters, same run repeated)."""
    passes = tracks * 5 * 2                             # (track, level, direction) triples
    flop = 11907.0 * passes + 5292.0 * iterations      # 441 px x (27 set-up + 12 per iteration) fp32 flop
    lds = 441.0 * 4 * 4 * iterations                   # 4 bilinear taps x 4 B per window pixel per iteration
    rows, cols = shape
    pyr = 4.0 * rows * cols * sum(0.25 ** l for l in range(5))
    s = total_ms * 1e-3
    # (the library launches the build for 5 wavefronts per SIMD for a group of loops, the one for 3 for a single loop)
    name = "klt_track_kernel<21, 5>" if group else "klt_track_kernel<21, 3>"
    return {"kernel": name, "launches": launches, "tracks": tracks, "tracks_per_launch": tracks / max(launches, 1),
            "kernel_ms_total": total_ms, "kernel_ms_mean": total_ms / max(launches, 1), "kernel_ms_longest": longest_ms,
            "tracker_us_per_frame": 1e3 * total_ms / max(frames_of_run, 1),
            "mean_iterations_per_pass": iterations / max(passes, 1), "algorithmic_flop": flop, "algorithmic_lds_bytes": lds,
            "compulsory_hbm_bytes_per_frame": 2 * pyr + 40.0 * tracks / max(frames_of_run, 1),
            "achieved_lds_tbs": lds / s / 1e12, "peak_lds_tbs": LDS_PEAK_TBS, "frac_lds": lds / s / 1e12 / LDS_PEAK_TBS,
            "achieved_tflops": flop / s / 1e12, "peak_tflops": FP32_VALU_PEAK_TFLOPS,
            "frac_valu": flop / s / 1e12 / FP32_VALU_PEAK_TFLOPS}


def tracker_of_loop(vo, torch, ctx, scene, frames, poses, steps, warmup, seed):
    """The headline loop twice more, untimed: once with the tracker launches bracketed by timing events, once with
    the tracker's work counters on -> the roofline entry of the loop's OWN tracker launches (two per frame: the
    kept landmarks', and the appended landmarks' + candidates')."""
    import ctypes as C
    lib = vo._lib.load()
    tot, longest, launches = C.c_double(0), C.c_double(0), C.c_int(0)
    tr, it = C.c_longlong(0), C.c_longlong(0)
    nosync = lambda: torch.cuda.synchronize()   # noqa: E731
    nogather = lambda pipe, seconds: []          # noqa: E731
    ctx.set_option("klt_time", 1)
    bench_frames(vo, torch, ctx, scene, frames, poses, steps, warmup, seed, nosync, nogather)
    ctx.check(lib.vohip_klt_time(ctx.handle, C.byref(tot), C.byref(longest), C.byref(launches), 1))
    ctx.set_option("klt_time", 0)
    ctx.set_option("klt_stats", 1)
    ctx.check(lib.vohip_klt_stats(ctx.handle, C.byref(tr), C.byref(it), 1))
    bench_frames(vo, torch, ctx, scene, frames, poses, steps, warmup, seed, nosync, nogather)
    ctx.check(lib.vohip_klt_stats(ctx.handle, C.byref(tr), C.byref(it), 1))
    ctx.set_option("klt_stats", 0)
    return tracker_roofline(tot.value, longest.value, launches.value, tr.value, it.value, frames[0].shape,
                            frames_of_run=warmup + steps)


def bench_main_eval(vo, ctx, frames, poses, concurrent=1, group=0, feats=("Harris", "MinEigen", "FAST"), repetition=4):
    """The batch evaluation of main_eval.m:20-23 on one GPU: a grid of 3 feature variants (Harris, MinEigen, FAST) x 4
    repetitions = 12 full runs (bootstrap on the first three frames, continuous operation on the rest,
    quantitative_eval, result record) of one KITTI-shape sequence whose images live in host memory; `concurrent`
    runs in flight at a time (a thread and a context each), or -- group -- the Harris and MinEigen runs as groups
    of that many loops in lockstep.  Returns runs/s and the frames/s of the grid as a whole (continuous-operation
    frames / wall time, bootstraps and evaluations included).  feats / repetition: another grid of the same size,
    e.g. main_eval.m:19's own four variants with SURF, the reference's default, among them."""
    seq = dict(name="kitti", images=frames, K=vo.synth.K_KITTI,
               ground_truth=np.array([[T[0, 3], T[2, 3]] for T in poses[:len(frames)]]))
    short = dict(seq, images=frames[:12])
    vo.main_eval([short], feats=("Harris", "FAST") if "SURF" not in feats else feats,
                 repetition=2 if group and concurrent > 1 else 1, seed=10, ctx=ctx,
                 concurrent=concurrent, group=min(group, 2) if concurrent > 1 else group)   # code objects, staging buffers
    t0 = time.perf_counter()
    records = vo.main_eval([seq], feats=feats, repetition=repetition, seed=10, ctx=ctx, concurrent=concurrent, group=group)
    dt = time.perf_counter() - t0
    n_frames = sum(int(r["frame_ct"]) for r in records)
    return {"runs": len(records), "runs_per_s": len(records) / dt, "frames_per_s": n_frames / dt,
            "frames_per_run": n_frames // len(records), "mean_fps_ct_of_a_run": float(np.mean([r["fps_ct"] for r in records])),
            "errs_columns": [int(r["errs"].shape[1]) for r in records]}


def bench_parking(vo, torch, ctx, steps, warmup, bundle_adjust):
    """The reference's parking configuration (main_demo.m:100-112; with bundle_adjust main_BA.m:197-198: window of
    5 keyframes, every 2nd frame a keyframe) on a rendered 640 x 480 sequence seen through the parking calibration
    the reference holds (at half resolution), frames resident in HBM; returns frames/s and the mean counts."""
    K, W, H = vo.synth.K_PARKING, vo.synth.PARKING_W, vo.synth.PARKING_H
    scene = vo.synth.CorridorScene(K=K, width=W, height=H, half_width=5.0, roof=4.0)
    poses = vo.synth.corridor_trajectory(warmup + steps + 2, step=0.3)
    frames = [scene.render(T) for T in poses]
    dev = torch.device("cuda", torch.cuda.current_device())
    d_frames = [torch.from_numpy(f).to(dev) for f in frames]
    _, kp = vo.genKeypoints(frames[0], "Harris", ["MinQuality", 2e-4], ctx=ctx)
    xy = kp.Location.astype(np.float64)
    xy = xy[thin_by_grid(xy, kp.Metric)]
    hp = vo.default_hyper_paras("parking", is_BA=bundle_adjust)
    pipe = vo.VOPipelineDevice(K, d_frames[0], scene.backproject(poses[0], xy), xy, poses[0], hyper_paras=hp,
                               rng=np.random.default_rng(7), ctx=ctx, window_size=5, keyframe_d=1)
    for i in range(1, warmup + 1):
        pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup + 1, warmup + steps + 1):
        pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"frames_per_s": steps / dt, "tracked_landmarks": int(np.mean(pipe.num_X[-steps:])),
           "candidates": int(np.mean(pipe.num_C[-steps:])), "ba_runs": pipe.ba_runs,
           "final_position_error_m": float(np.linalg.norm(pipe.est_trans[-1] - poses[warmup + steps][:3, 3]))}
    pipe.close()
    return out


def bench_ransac(vo, torch, ctx, n, h, steps, warmup, seed):
    """RANSAC stress: h P3P hypotheses scored against n correspondences per step."""
    lib = vo._lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    kp_rc, X, _, _, _ = vo.synth.pnp_problem(n, seed=seed)
    samples = vo.synth.sample_triples(n, h, seed=42 + seed)
    d_kp = torch.from_numpy(np.ascontiguousarray(kp_rc.T)).to(dev)
    d_X = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    d_s = torch.from_numpy(samples).to(dev)
    d_c = torch.zeros((h, 4), dtype=torch.int32, device=dev)
    Kp = vo._lib.dptr(vo._lib.colmajor(vo.synth.K_KITTI))

    def step():
        ctx.check(lib.vohip_ransac_p3p_score_dev(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), n, Kp, d_s.data_ptr(),
                                                  h, d_c.data_ptr()))

    def count_only():
        ctx.check(lib.vohip_ransac_p3p_count_dev(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), n, h, d_c.data_ptr()))

    # The clocks need a few milliseconds of load to settle (a 2 ms burst of fp64 FMAs runs at ~50-60
    # TFLOP/s, a sustained one at ~72), so both measurements run long enough to be steady-state.
    steps = max(steps, 200) if h <= 4096 else steps
    for _ in range(max(warmup, 50) if h <= 4096 else warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    hyps_per_s = h * steps / (time.perf_counter() - t0)
    # scoring kernel alone, HIP events on the stream it is launched on
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = steps
    for _ in range(50):
        count_only()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        count_only()
    ev1.record()
    torch.cuda.synchronize()
    score_ms = ev0.elapsed_time(ev1) / reps
    return hyps_per_s, score_ms


def bench_klt(vo, torch, ctx, d_frames, xy, n_points, reps=20):
    """The tracker kernel alone (21x21, 5 levels, 20 iterations, forward-backward) on two consecutive
    frames: kernel time from HIP events on its stream, work from the library's own counters."""
    import ctypes as C
    lib = vo._lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    reps_xy = int(np.ceil(n_points / len(xy)))
    pts = np.concatenate([xy + 0.37 * k for k in range(reps_xy)])[:n_points]
    p0, p1 = vo.Pyramid(d_frames[0], 5, ctx=ctx), vo.Pyramid(d_frames[1], 5, ctx=ctx)
    d_pts = torch.from_numpy(np.ascontiguousarray(pts.T)).to(dev)
    d_out = torch.zeros_like(d_pts)
    d_valid = torch.zeros(n_points, dtype=torch.uint8, device=dev)

    def run():
        ctx.check(lib.vohip_klt_track_dev(ctx.handle, p0.slot, p1.slot, d_pts.data_ptr(), n_points, 21, 20, 6.0,
                                          d_out.data_ptr(), d_valid.data_ptr(), None))

    run()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        run()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    # one more launch with the work counters on (they cost two atomics per track, so not while timing)
    tr, it = C.c_longlong(0), C.c_longlong(0)
    ctx.set_option("klt_stats", 1)
    ctx.check(lib.vohip_klt_stats(ctx.handle, C.byref(tr), C.byref(it), 1))
    run()
    ctx.check(lib.vohip_klt_stats(ctx.handle, C.byref(tr), C.byref(it), 1))
    ctx.set_option("klt_stats", 0)
    p0.release()
    p1.release()
    passes = tr.value * 5 * 2                          # (track, level, direction) triples per launch
    iters = it.value
    flop = 11907.0 * passes + 5292.0 * iters          # 441 px x (27 set-up + 12 per iteration) fp32 flop
    lds = 441.0 * 4 * 4 * iters                       # 4 bilinear taps x 4 B per window pixel per iteration
    # compulsory HBM bytes of a launch: both pyramids once (float, 4/3 of the image each) + 40 B per track
    rows, cols = d_frames[0].shape
    hbm = 2 * 4.0 * rows * cols * sum(0.25 ** l for l in range(5)) + 40.0 * n_points
    return dict(kernel="klt_track_kernel<21, 5>" if n_points >= 8192 else "klt_track_kernel<21, 3>", tracks=n_points, kernel_ms=ms, mean_iterations_per_pass=iters / max(passes, 1),
                algorithmic_flop=flop, algorithmic_lds_bytes=lds, compulsory_hbm_bytes=hbm,
                valid_fraction=float(d_valid.float().mean().item()), achieved_tflops=flop / (ms * 1e-3) / 1e12,
                peak_tflops=FP32_VALU_PEAK_TFLOPS, frac_valu=flop / (ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS,
                achieved_lds_tbs=lds / (ms * 1e-3) / 1e12, peak_lds_tbs=LDS_PEAK_TBS,
                frac_lds=lds / (ms * 1e-3) / 1e12 / LDS_PEAK_TBS)


def bench_image_kernels(vo, torch, ctx, d_frame, reps=50, d_frames=None):
    """The two per-image stages that stream the frame: pyramid build (5 levels) and corner detection
    (Harris response, non-maximum suppression, ordered emit), one frame per launch sequence as the loop
    runs them.  Algorithmic HBM bytes: pyramid = the uint8 image in, every fp32 level written once and read
    once for the next level; detection = the image in and 8 bytes per corner out."""
    lib = vo._lib.load()
    dev = d_frame.device
    rows, cols = d_frame.shape
    cap = 16384
    d_xy = torch.zeros(2 * cap, dtype=torch.float32, device=dev)
    d_metric = torch.zeros(cap, dtype=torch.float32, device=dev)
    d_count = torch.zeros(1, dtype=torch.int32, device=dev)
    pyr = vo.Pyramid(d_frame, 5, ctx=ctx)

    def build():
        ctx.check(lib.vohip_pyramid_build_dev(ctx.handle, pyr.slot, d_frame.data_ptr(), rows, cols, 0, 5))

    def detect():
        ctx.check(lib.vohip_detect_corners_dev(ctx.handle, d_frame.data_ptr(), rows, cols, 0, 2e-4, 5, 16, 0,
                                               d_xy.data_ptr(), d_metric.data_ptr(), cap, d_count.data_ptr()))

    def detect_fast():   # main_eval.m:172-173
        ctx.check(lib.vohip_detect_fast_dev(ctx.handle, d_frame.data_ptr(), rows, cols, 0, 0.11, 0.1, 16,
                                            d_xy.data_ptr(), d_metric.data_ptr(), cap, d_count.data_ptr()))

    def detect_brisk():   # main_eval.m:170-171
        ctx.check(lib.vohip_detect_brisk_dev(ctx.handle, d_frame.data_ptr(), rows, cols, 0, 0.2, 2e-3, 4, 16, d_xy.data_ptr(),
                                             d_metric.data_ptr(), None, cap, d_count.data_ptr()))

    def detect_surf():   # main_eval.m:166-167
        ctx.check(lib.vohip_detect_surf_dev(ctx.handle, d_frame.data_ptr(), rows, cols, 0, 200.0, 16, d_xy.data_ptr(),
                                            d_metric.data_ptr(), None, cap, d_count.data_ptr()))

    out = {}
    for name, fn in (("pyramid_build", build), ("fast_detection", detect_fast), ("surf_detection", detect_surf),
                     ("brisk_detection", detect_brisk), ("corner_detection", detect)):
        fn()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        out[name] = ev0.elapsed_time(ev1) / reps
        if name == "fast_detection":
            n_fast = min(int(d_count.item()), cap)
        if name == "surf_detection":
            n_surf = min(int(d_count.item()), cap)
        if name == "brisk_detection":
            n_brisk = min(int(d_count.item()), cap)
    px = float(rows * cols)
    levels = sum(0.25 ** l for l in range(5))
    n_corners = min(int(d_count.item()), cap)
    bytes_pyr = px * (1.0 + 4.0 * levels + 4.0 * (levels - 0.25 ** 4))
    bytes_det = px + 8.0 * n_corners
    pyr.release()
    # The same two stages over a stored sequence of 256 frames in ONE launch sequence (vohip_pyramid_batch_dev,
    # vohip_detect_corners_batch_dev: image = a grid dimension), where they stream instead of waiting on the
    # latency of a 0.47 MB image: this is where their HBM fraction can be read.
    nb = 256
    d_batch = torch.stack([d_frames[k % len(d_frames)] for k in range(nb)]) if d_frames else d_frame.repeat(nb, 1, 1)
    per = lib.vohip_pyramid_floats(rows, cols, 0, 5)
    d_pyr = torch.empty((nb, per), dtype=torch.float32, device=dev)
    b_xy = torch.zeros((nb, 2 * cap), dtype=torch.float32, device=dev)
    b_metric = torch.zeros((nb, cap), dtype=torch.float32, device=dev)
    b_counts = torch.zeros(nb, dtype=torch.int32, device=dev)

    def build_batch():
        ctx.check(lib.vohip_pyramid_batch_dev(ctx.handle, d_batch.data_ptr(), nb, rows, cols, 0, 5, d_pyr.data_ptr()))

    def detect_batch():
        ctx.check(lib.vohip_detect_corners_batch_dev(ctx.handle, d_batch.data_ptr(), nb, rows, cols, 0, 2e-4, 5, 16, 0,
                                                     b_xy.data_ptr(), b_metric.data_ptr(), cap, b_counts.data_ptr()))

    def detect_brisk_batch():
        ctx.check(lib.vohip_detect_brisk_batch_dev(ctx.handle, d_batch.data_ptr(), nb, rows, cols, 0, 0.2, 2e-3, 4, 16,
                                                   b_xy.data_ptr(), b_metric.data_ptr(), cap, b_counts.data_ptr()))

    def detect_surf_batch():
        ctx.check(lib.vohip_detect_surf_batch_dev(ctx.handle, d_batch.data_ptr(), nb, rows, cols, 0, 200.0, 16,
                                                  b_xy.data_ptr(), b_metric.data_ptr(), cap, b_counts.data_ptr()))

    def detect_fast_batch():   # main_eval.m:172-173
        ctx.check(lib.vohip_detect_fast_batch_dev(ctx.handle, d_batch.data_ptr(), nb, rows, cols, 0, 0.11, 0.1, 16,
                                                  b_xy.data_ptr(), b_metric.data_ptr(), cap, b_counts.data_ptr()))

    batched = {}
    for name, fn in (("pyramid_build", build_batch), ("surf_detection", detect_surf_batch),
                     ("brisk_detection", detect_brisk_batch), ("fast_detection", detect_fast_batch),
                     ("corner_detection", detect_batch)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        batched[name] = ev0.elapsed_time(ev1) / 5
        if name == "surf_detection":
            surf_b = float(torch.clamp(b_counts, max=cap).sum().item())
        if name == "brisk_detection":
            brisk_b = float(torch.clamp(b_counts, max=cap).sum().item())
        if name == "fast_detection":
            fast_b = float(torch.clamp(b_counts, max=cap).sum().item())
    corners_b = float(torch.clamp(b_counts, max=cap).sum().item())
    bytes_det_b = px * nb + 8.0 * corners_b
    # SURF per image: the image in, the integral image (4 bytes per pixel) written by the row pass, read and
    # written by the column pass and read by the response pass (32 lookups per response come out of L2), the 8
    # response maps of octaves 1 and 2 (4 (1/4 + 1/16) floats per pixel) written once and read once (octave 0 is
    # tested from LDS and never stored), 16 bytes per point out
    maps_px = 4.0 * (1.0 + 0.25 + 0.0625)
    bytes_surf = px * (1.0 + 4.0 * 4.0 + 2.0 * 4.0 * 4.0 * (0.25 + 0.0625))
    # BRISK per image: the image read by the two resamplings and the score pass, 1.0 byte per pixel of coarser layers
    # written and read, 2.2 + 1 bytes per pixel of scores written once and read once, 16 bytes per keypoint out
    bytes_brisk = px * (3.0 + 2.0 * 1.0 + 2.0 * 3.2)
    batch_out = {"frames": nb,
                 "brisk_detection": {"ms": batched["brisk_detection"], "points": brisk_b,
                                     "algorithmic_bytes": bytes_brisk * nb + 16.0 * brisk_b,
                                     "achieved_gbs": (bytes_brisk * nb + 16.0 * brisk_b) / (batched["brisk_detection"] * 1e-3) / 1e9,
                                     "peak_gbs": 8000.0,
                                     "frac": (bytes_brisk * nb + 16.0 * brisk_b) / (batched["brisk_detection"] * 1e-3) / 1e9 / 8000.0,
                                     "note": "ten launches serve the batch (seven resamplings, scores, extrema, placement); the "
                                             "score pass (about 180 integer operations per pixel of 2.2 images) bounds it"},
                 "surf_detection": {"ms": batched["surf_detection"], "points": surf_b,
                                    "algorithmic_bytes": bytes_surf * nb + 16.0 * surf_b,
                                    "achieved_gbs": (bytes_surf * nb + 16.0 * surf_b) / (batched["surf_detection"] * 1e-3) / 1e9,
                                    "peak_gbs": 8000.0,
                                    "frac": (bytes_surf * nb + 16.0 * surf_b) / (batched["surf_detection"] * 1e-3) / 1e9 / 8000.0,
                                    "lookups_per_s": 32.0 * maps_px * px * nb / (batched["surf_detection"] * 1e-3),
                                    "note": "six launches serve the batch; the response pass makes 32 four-byte lookups of the "
                                            "integral image per response (LDS tiles for octave 0, the L2-resident 1.9 MB "
                                            "integral image for octaves 1 and 2), which is what it is bound by"},
                 "pyramid_build": {"ms": batched["pyramid_build"], "algorithmic_bytes": bytes_pyr * nb,
                                   "achieved_gbs": bytes_pyr * nb / (batched["pyramid_build"] * 1e-3) / 1e9, "peak_gbs": 8000.0,
                                   "frac": bytes_pyr * nb / (batched["pyramid_build"] * 1e-3) / 1e9 / 8000.0,
                                   "compulsory_bytes": px * (1.0 + 4.0 * levels) * nb,
                                   "frac_of_compulsory": px * (1.0 + 4.0 * levels) * nb / (batched["pyramid_build"] * 1e-3) / 1e9 / 8000.0,
                                   "note": "algorithmic_bytes counts every level written once and read once for the next one; "
                                           "compulsory_bytes = the image in and every level written once (levels 0 and 1 come "
                                           "out of one pass over the image, so level 0 is not read back)"},
                 "fast_detection": {"ms": batched["fast_detection"], "corners": fast_b,
                                    "algorithmic_bytes": px * nb + 8.0 * fast_b,
                                    "achieved_gbs": (px * nb + 8.0 * fast_b) / (batched["fast_detection"] * 1e-3) / 1e9,
                                    "peak_gbs": 8000.0,
                                    "frac": (px * nb + 8.0 * fast_b) / (batched["fast_detection"] * 1e-3) / 1e9 / 8000.0,
                                    "note": "about 180 integer operations per pixel (the segment-test score): bound by its "
                                            "arithmetic, not by the byte per pixel it streams"},
                 "corner_detection": {"ms": batched["corner_detection"], "corners": corners_b, "algorithmic_bytes": bytes_det_b,
                                      "achieved_gbs": bytes_det_b / (batched["corner_detection"] * 1e-3) / 1e9,
                                      "peak_gbs": 8000.0,
                                      "frac": bytes_det_b / (batched["corner_detection"] * 1e-3) / 1e9 / 8000.0,
                                      "achieved_tflops_fp32": 80.0 * px * nb / (batched["corner_detection"] * 1e-3) / 1e12,
                                      "note": "about 80 fp32 flop per pixel: the detector is bound by its arithmetic and LDS "
                                              "traffic, not by the 1 byte per pixel it streams"}}
    del d_batch, d_pyr, b_xy, b_metric, b_counts
    return {"batched_256_frames": batch_out, "note": "per-frame launch sequences of %dx%d; at 0.47 MB per frame these stages are launch- and "
                    "latency-bound, far from the HBM roofline" % (cols, rows),
            "pyramid_build": {"ms": out["pyramid_build"], "algorithmic_bytes": bytes_pyr,
                              "achieved_gbs": bytes_pyr / (out["pyramid_build"] * 1e-3) / 1e9, "peak_gbs": 8000.0},
            "corner_detection": {"ms": out["corner_detection"], "corners": n_corners, "algorithmic_bytes": bytes_det,
                                 "achieved_gbs": bytes_det / (out["corner_detection"] * 1e-3) / 1e9,
                                 "peak_gbs": 8000.0},
            "brisk_detection": {"ms": out["brisk_detection"], "points": n_brisk, "algorithmic_bytes": bytes_brisk + 16.0 * n_brisk,
                                "achieved_gbs": (bytes_brisk + 16.0 * n_brisk) / (out["brisk_detection"] * 1e-3) / 1e9,
                                "peak_gbs": 8000.0},
            "surf_detection": {"ms": out["surf_detection"], "points": n_surf, "algorithmic_bytes": bytes_surf + 16.0 * n_surf,
                               "achieved_gbs": (bytes_surf + 16.0 * n_surf) / (out["surf_detection"] * 1e-3) / 1e9,
                               "peak_gbs": 8000.0},
            "fast_detection": {"ms": out["fast_detection"], "corners": n_fast, "algorithmic_bytes": px + 8.0 * n_fast,
                               "achieved_gbs": (px + 8.0 * n_fast) / (out["fast_detection"] * 1e-3) / 1e9,
                               "peak_gbs": 8000.0}}


def bench_ba(vo, ctx, with_cpu):
    """The sliding-window bundle adjustment of main_BA.m (runBA: F = 5 keyframes, m landmarks, 60 % visible,
    0.5 px noise; one call = one whole Levenberg-Marquardt run from the perturbed state, host buffers in and
    out as the reference calls it), next to the CPU restatement of the same run."""
    out = {}
    for m in (300, 1000):
        s0, obs, _ = vo.synth.ba_problem(F=5, m=m, seed=9)
        _, info = vo.runBA(s0, obs, vo.synth.K_KITTI, return_stats=True, ctx=ctx)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            vo.runBA(s0, obs, vo.synth.K_KITTI, ctx=ctx)
        ms = (time.perf_counter() - t0) / reps * 1e3
        n_obs = int((len(obs) - 2 - 5) // 3)
        rec = {"ms_per_run": ms, "lm_iterations": info["iterations"], "observations": n_obs,
               "cost": [info["cost0"], info["cost"]]}
        if with_cpu:
            from oracle import vo_oracle
            vo_oracle.build()
            t0 = time.perf_counter()
            vo_oracle.run_ba(s0, obs, vo.synth.K_KITTI)
            rec["cpu_restatement_ms_per_run"] = (time.perf_counter() - t0) * 1e3
        out["m=%d" % m] = rec
    return out


def traffic_from_profile(kernel, prefer=None):
    """HBM bytes per launch of `kernel` as the committed rocprofv3 counter passes saw them (newest
    profiles/*_summary.csv that lists it; with `prefer`, files whose name holds that word first): FETCH_SIZE +
    WRITE_SIZE, which rocprofv3 reports in KB, as read (the kernel's loads are 8 bytes per lane, for which the
    counter is not calibrated).  None without a profile."""
    import csv
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.csv")), reverse=True)
    if prefer:
        paths = [q for q in paths if prefer in os.path.basename(q)] + [q for q in paths if prefer not in os.path.basename(q)]
    for path in paths:
        try:
            for row in csv.DictReader(open(path, newline="")):
                if row.get("kernel") == kernel and row.get("FETCH_SIZE") and row.get("WRITE_SIZE"):
                    return {"bytes": (float(row["FETCH_SIZE"]) + float(row["WRITE_SIZE"])) * 1024.0,
                            "source": os.path.relpath(path, ROOT)}
        except (OSError, ValueError, KeyError):
            continue
    return None


def cpu_baseline_frames(vo, scene, frames, poses, X0, xy0, budget_s=12.0, threads=0):
    """The CPU restatement of the same loop on the first frames of the same sequence.  threads: for the tracker's
    loop over the points, the one parallel part (0: all cores)."""
    from oracle import pipeline_oracle as po
    from oracle import vo_oracle
    vo_oracle.build()
    vo_oracle.set_threads(threads)
    hp = vo.default_hyper_paras("kitti", is_BA=False)
    hp.r_discard_redundant = KITTI_EXCLUSION_RADIUS
    pipe = po.OraclePipeline(vo.synth.K_KITTI, frames[0], X0, xy0, poses[0], hp, 3, 2, np.random.default_rng(0),
                             vo.draw_triples, vo.draw_octets)
    pipe.process_frame(frames[1])  # first frame builds up the candidate set; not timed
    done, t0 = 0, time.perf_counter()
    for i in range(2, len(frames)):
        pipe.process_frame(frames[i])
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done / dt, done, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=8192)
    ap.add_argument("--hypotheses", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--thin-cell", type=float, default=KITTI_THIN_CELL, help="thinning cell of the first frame's corners, px")
    ap.add_argument("--exclusion-radius", type=float, default=KITTI_EXCLUSION_RADIUS,
                    help="r_discard_redundant of the KITTI-shape workload, px (the reference's value is 10)")
    ap.add_argument("--only-headline", action="store_true",
                    help="the headline loop and the roofline entry of its own tracker launches, nothing else")
    ap.add_argument("--only-group", type=int, default=0, metavar="S",
                    help="the headline loop and a group of S loops in lockstep, nothing else (for profiles)")
    args = ap.parse_args()
    globals().update(KITTI_THIN_CELL=args.thin_cell, KITTI_EXCLUSION_RADIUS=args.exclusion_radius)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # VOHIP_BENCH_BACKEND=gloo lets several ranks rehearse the multi-process path on one GPU
    backend = os.environ.get("VOHIP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = vo.Context(local_rank, stream=stream.cuda_stream)
        gather_s = []

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        def gather(pipe, seconds):
            t0 = time.perf_counter()
            et, er = pipe.trajectory()
            rec = vo.batch.pack_record(rank, seconds, et, er, pipe.num_X, pipe.num_C, pipe.num_new,
                                       timing=dict(toc_ct=seconds, frame_ct=len(pipe.num_X) - 1,
                                                   fps_ct=(len(pipe.num_X) - 1) / seconds))
            out = vo.batch.gather_records([rec], world, device=coll_device if (world > 1 and backend == "nccl") else None)
            gather_s.append(time.perf_counter() - t0)
            return out

        grid_group = 4   # (N > 1) work items of the run grid a rank advances together in the grid leg
        n_frames = max(args.warmup + args.steps + 2, 8)   # (the priming run below needs 8)
        # with the CPU baseline leg the sequence is longer, so that its sample is about 10 s of CPU work; the group
        # legs stagger their sequences by a frame each
        quick = args.only_headline or args.only_group
        n_seq = n_frames + (max(args.only_group, 1) if quick else 32)
        if world > 1:
            n_seq = n_frames + grid_group
        elif not args.no_cpu_baseline and not quick:
            n_seq = max(n_seq, 212)
        # work item of this rank (the batch grid main_eval.m:22-23 shards): its own rendered sequence and seed
        item = vo.batch.shard(list(range(world)), rank, world)[0]
        scene, frames_all, poses_all = make_sequence(vo, n_seq, item)
        frames, poses = frames_all[:n_frames], poses_all[:n_frames]
        # set-up, not steps: a throw-away run of a few frames loads every kernel's code object and maps the
        # staging buffers, so that the W warm-up steps below start from a process that has run the path once;
        # its gather is the first use of the collective (communicator set-up, buffers), which is not the job's cost
        bench_frames(vo, torch, ctx, scene, frames, poses, 4, 2, 99, barrier, gather)
        gather_s.clear()
        elapsed, info = bench_frames(vo, torch, ctx, scene, frames, poses, args.steps, args.warmup, 100 + item, barrier,
                                     gather)
        if world > 1:
            t = torch.tensor([elapsed, gather_s[-1]], dtype=torch.float64, device=coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, gather_max = float(t[0].item()), float(t[1].item())
        else:
            gather_max = gather_s[-1]
        fps = world * args.steps / elapsed
        extra = {}
        nosync = lambda: torch.cuda.synchronize()   # noqa: E731
        nogather = lambda pipe, seconds: []          # noqa: E731
        if world > 1:
            # The unit the reference's batch mode shards (main_eval.m:22-23) as this build runs it on a GPU: every
            # rank advances `grid_group` work items of the run grid together, as a group of loops in lockstep (own
            # seed and start frame each, the rank's own rendered corridor); no exchange while they run; the records
            # are gathered afterwards (timed apart).
            barrier()
            grid = bench_group(vo, torch, ctx, scene, frames_all, poses_all, args.steps, args.warmup, grid_group,
                               seed=1000 + 100 * rank)
            t = torch.tensor([grid["ms_per_step"]], dtype=torch.float64, device=coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            extra["run_grid_groups_of_%d" % grid_group] = {
                "frames_per_s": world * grid_group * 1e3 / float(t.item()), "ms_per_step_slowest_rank": float(t.item()),
                "items_per_rank": grid_group, "frames_per_item": args.steps,
                "note": "scaling of the batch mode proper: world x %d runs, each rank its group of loops in lockstep; "
                        "divide by the same field of the 1-GPU line (frames_per_s_group_of_4)" % grid_group}
        if world == 1:
            # the same loop with its images in pinned host memory: every frame's 466 KB upload is part of the frame
            # (frames of the loop / wall time of the loop, image decode excluded, upload included)
            el, _ = bench_frames(vo, torch, ctx, scene, frames, poses, args.steps, args.warmup, 100, nosync, nogather,
                                 resident=False)
            extra["frames_per_s_pcie_inclusive"] = args.steps / el
        if world == 1 and args.only_group:
            extra["frames_per_s_group_of_%d" % args.only_group] = bench_group(
                vo, torch, ctx, scene, frames_all, poses_all, args.steps, args.warmup, args.only_group, measure_tracker=True)
        if world == 1 and not quick:
            el, _ = bench_frames(vo, torch, ctx, scene, frames, poses, args.steps, args.warmup, 100, nosync, nogather,
                                 ahead=False)
            extra["frames_per_s_without_lookahead"] = args.steps / el
            # the headline loop with the other detectors of main_eval.m:165-174 (SURF is the reference's default,
            # main_demo.m:25): only the detection on the side stream and the candidates it feeds differ
            variants = {}
            for feat in (("SURF", ["MetricThreshold", 200]), ("BRISK", ["MinQuality", 2e-3]), ("FAST", ["MinContrast", 0.11])):
                el, inf = bench_frames(vo, torch, ctx, scene, frames, poses, args.steps, args.warmup, 100, nosync, nogather,
                                       feature=feat)
                variants[feat[0]] = {"frames_per_s": args.steps / el, "options": feat[1],
                                     "tracked_landmarks": inf["n_landmarks"], "candidates": inf["n_candidates"],
                                     "final_position_error_m": inf["final_position_error_m"]}
            extra["frames_per_s_feature_variants"] = variants
            # main_BA.m's configuration on the same KITTI-shape sequence: bundle adjustment over a window of 5
            # keyframes, every 2nd frame a keyframe (runBA and the normal-frame re-refinements on the GPU, the
            # window bookkeeping on the host, one copy of the landmarks each way)
            el, inf = bench_frames(vo, torch, ctx, scene, frames, poses, args.steps, args.warmup, 100, nosync, nogather,
                                   bundle_adjust=True)
            extra["frames_per_s_with_bundle_adjustment"] = {
                "value": args.steps / el, "ba_runs_in_timed_and_warmup_frames": inf["ba_runs"],
                "tracked_landmarks": inf["n_landmarks"], "candidates": inf["n_candidates"],
                "reference_published_fps": MATLAB_PARKING_BA_FPS,
                "note": "KITTI shape; the reference's only published figure with bundle adjustment is parking, MATLAB"}
            # the reference's parking configuration (BASELINE config 1 and, with bundle adjustment, config 3)
            steps_p = min(args.steps, 60)
            extra["parking_640x480"] = {
                "note": "a proxy of ds = 2: the reference's data/parking/K.txt is absent, so the rendered 640 x 480 corridor is "
                        "seen through data/epfl_parking/calibration/K.txt at half resolution",
                "no_ba": dict(bench_parking(vo, torch, ctx, steps_p, args.warmup, False),
                              reference_published_fps=MATLAB_PARKING_FPS),
                "with_ba_window5_every_2nd_frame": dict(bench_parking(vo, torch, ctx, steps_p, args.warmup, True),
                                                        reference_published_fps=MATLAB_PARKING_BA_FPS)}
            # the whole loop at the stress sizes (the kernel-level stress figures follow below)
            steps_s = min(args.steps, 40)
            el_s, info_s = bench_frames(vo, torch, ctx, scene, frames, poses, steps_s, args.warmup, 100, nosync,
                                        nogather, stress_hypotheses=args.hypotheses)
            extra["frame_loop_stress_config"] = {
                "workload": "every corner of the frames (Harris MinQuality %g: about 8 250 per frame, all of the first "
                            "frame's start as landmarks), exclusion radius %g px, %d P3P hypotheses solved and scored "
                            "per frame; the rendered frames hold no more corners than that, so 8 192 + 8 192 distinct "
                            "tracks do not exist in them (the tracker and RANSAC kernels alone are measured at 8 192 below)"
                            % (STRESS_MIN_QUALITY, STRESS_EXCLUSION_RADIUS, args.hypotheses),
                "frames_per_s": steps_s / el_s, "tracked_landmarks": info_s["n_landmarks"],
                "candidates": info_s["n_candidates"], "final_position_error_m": info_s["final_position_error_m"]}
            # Independent sequences on one GPU (the runs of main_eval.m:22-23).  As a group of loops in lockstep:
            # every kernel of a frame launched once for all of them ...
            for S in (4, 8, 16):
                extra["frames_per_s_group_of_%d" % S] = bench_group(vo, torch, ctx, scene, frames_all, poses_all, args.steps,
                                                                    args.warmup, S, measure_tracker=S == 16)
            # ... two such groups on two contexts, begun in turn by the one host thread ...
            for S in (16, 32):
                extra["frames_per_s_2_groups_of_%d_in_turn" % (S // 2)] = bench_group(
                    vo, torch, ctx, scene, frames_all, poses_all, args.steps, args.warmup, S, groups=2)
            # ... and as four loops on four host threads, a context and stream set each (what the group replaces)
            steps_c = min(args.steps, 100)
            extra["frames_per_s_4_sequences_at_once"] = bench_frames_concurrent(vo, torch, scene, frames_all, poses_all,
                                                                                steps_c, args.warmup, 4)
            extra["frames_per_sequence_in_that_figure"] = steps_c
            # BASELINE's batch configuration on one GPU (with N GPUs the grid is dealt to the ranks longest first)
            extra["main_eval_grid_one_gpu"] = {
                "workload": "main_eval.m:20-23: 3 feature variants (Harris, MinEigen, FAST) x 4 repetitions of one KITTI-shape "
                            "sequence of %d frames, images in host memory; bootstrap + loop + evaluation per run" % len(frames),
                "one_run_at_a_time": bench_main_eval(vo, ctx, frames, poses),
                "four_runs_in_flight": bench_main_eval(vo, ctx, frames, poses, concurrent=4),
                "groups_of_8_loops_in_lockstep": bench_main_eval(vo, ctx, frames, poses, group=8),
                "two_groups_of_6_in_flight": bench_main_eval(vo, ctx, frames, poses, concurrent=2, group=6),
                "two_groups_of_6_in_flight_reference_grid": dict(
                    bench_main_eval(vo, ctx, frames, poses, concurrent=2, group=6,
                                    feats=("SURF", "Harris", "BRISK", "FAST"), repetition=3),
                    workload="main_eval.m:19's own four feature variants (SURF -- the reference's default --, Harris, BRISK, "
                             "FAST, with main_eval.m:165-174's options) x 3 repetitions: 12 runs")}

        klt_loop = klt_stress = image_kernels = ba_runs = None
        n, h = args.points, args.hypotheses
        hyps_per_s = score_ms = achieved = None
        if rank == 0:
            # the roofline entry of the headline loop's own tracker launches (untimed replays of the timed frames)
            klt_loop = tracker_of_loop(vo, torch, ctx, scene, frames, poses, args.steps, args.warmup, 100 + item)
        if rank == 0 and not quick:
            d2 = [torch.from_numpy(f).to(torch.device("cuda", local_rank)) for f in frames[:2]]
            _, xy0 = init_points(vo, ctx, scene, frames[0], poses[0])
            klt_stress = bench_klt(vo, torch, ctx, d2, xy0, 8192)
            d_many = [torch.from_numpy(f).to(torch.device("cuda", local_rank)) for f in frames[:32]]
            image_kernels = bench_image_kernels(vo, torch, ctx, d2[0], d_frames=d_many)
            del d_many
            ba_runs = bench_ba(vo, ctx, with_cpu=not args.no_cpu_baseline and world == 1)
            hyps_per_s, score_ms = bench_ransac(vo, torch, ctx, n, h, max(args.steps, 20), args.warmup, 100)
            # the other sizes of the micro-benchmark: correspondences x hypotheses per call, solve + score
            sweep = {}
            for nn in (2000, 8192):
                for hh in (128, 1024, 4096, 65536):
                    if (nn, hh) == (n, h):
                        sweep["n=%d,h=%d" % (nn, hh)] = {"hypotheses_per_s": hyps_per_s, "score_kernel_ms": score_ms,
                                                         "score_kernel_frac_fp64_valu": 116.0 * nn * hh / (score_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS}
                        continue
                    hs, ms = bench_ransac(vo, torch, ctx, nn, hh, 200 if hh <= 4096 else 50, 50 if hh <= 4096 else 10, 100)
                    sweep["n=%d,h=%d" % (nn, hh)] = {"hypotheses_per_s": hs, "score_kernel_ms": ms,
                                                     "score_kernel_frac_fp64_valu": 116.0 * nn * hh / (ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS}
            extra["ransac_sweep"] = sweep
            achieved = 116.0 * n * h / (score_ms * 1e-3) / 1e12

    if rank == 0:
        traffic = traffic_from_profile("p3p_score_kernel<16, 1024>") if (n, h) == (8192, 4096) else None
        klt_traffic = traffic_from_profile("klt_track_kernel<21, 3>", prefer="headline")
        out = {
            "metric": METRIC,
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": fps / world / MATLAB_KITTI_HARRIS_FPS if world == 1 else None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "KITTI-shape continuous operation: rendered corridor sequence 1241x376, Harris "
                                   "MinQuality 2e-4, 2 x KLT 21x21/5 levels, adaptive P3P RANSAC, pose refinement, "
                                   "F-matrix filter, triangulation, candidate filter, no BA; one step = one frame of ONE "
                                   "sequence, frames resident in HBM; "
                                   "the sequence is stored, so each call also hands over the following frame, whose pyramid "
                                   "and corners are prepared beside the current localisation (results identical); "
                                   "every rank runs its own work item of the batch grid (own rendered sequence, own "
                                   "random seed), records gathered with one all_gather inside the timed region",
                       "tracked_landmarks": info["n_landmarks"], "candidates": info["n_candidates"],
                       "first_frame_thinning_cell_px": KITTI_THIN_CELL, "candidate_exclusion_radius_px": KITTI_EXCLUSION_RADIUS,
                       "final_position_error_m": info["final_position_error_m"],
                       "host_enqueue_us_per_frame": info["host_enqueue_us_per_frame"],
                       "host_wait_us_per_frame": info["host_wait_us_per_frame"],
                       "gather_ms_inside_the_timed_region": 1e3 * gather_max,
                       "baseline_note": "vs_baseline divides by the reference's published 3.817 frames/s "
                                        "(MATLAB, Ryzen 7 5800H, real KITTI 05, Harris)"},
            # The kernel the headline is bound by: the tracker (two launches per frame: the landmarks the previous
            # frame kept, and the landmarks it appended + the candidates).  One wavefront per track; its hot loop
            # reads the search region from LDS and runs fp32 FMAs, so the launches are priced against the LDS
            # bandwidth (primary) and the fp32 vector peak; they are not HBM-bound (both pyramids, 5 MB, stay in
            # L2 / Infinity Cache).  Measured on the loop's OWN launches: durations from event pairs around every
            # tracker launch of an untimed replay of the timed frames, work from the kernel's counters.
            "roofline": {"kernel": "klt_track_kernel<21, 3> (the headline loop's own launches, two per frame)",
                         "bound": "lds",
                         "bound_note": "neither hbm nor mfma: the binding resource is LDS read bandwidth / fp32 VALU issue, and at "
                                       "these sizes the latency of the slowest wavefront (a lost track runs all 200 iterations); "
                                       "achieved = algorithmic LDS bytes (441 px x 4 taps x 4 B per iteration) / kernel time",
                         "achieved": klt_loop["achieved_lds_tbs"] * 1e3, "peak": LDS_PEAK_TBS * 1e3, "unit": "GB/s",
                         "frac": klt_loop["frac_lds"],
                         "frac_fp32_valu": klt_loop["frac_valu"], "achieved_tflops": klt_loop["achieved_tflops"],
                         "peak_tflops": FP32_VALU_PEAK_TFLOPS,
                         "kernel_ms": klt_loop["kernel_ms_mean"], "kernel_ms_longest": klt_loop["kernel_ms_longest"],
                         "launches": klt_loop["launches"], "tracks_per_launch": klt_loop["tracks_per_launch"],
                         "tracker_us_per_frame": klt_loop["tracker_us_per_frame"],
                         "mean_iterations_per_pass": klt_loop["mean_iterations_per_pass"],
                         "traffic": klt_traffic["bytes"] if klt_traffic else None,
                         "traffic_source": klt_traffic["source"] if klt_traffic else None,
                         "traffic_note": "FETCH_SIZE + WRITE_SIZE per launch as rocprofv3 reports them (KB x 1024) in the committed "
                                         "counter passes of `bench.py --only-headline`; the kernel's loads are 4 B per lane, "
                                         "for which the counter is not calibrated on gfx950",
                         "compulsory_hbm_bytes_per_frame": klt_loop["compulsory_hbm_bytes_per_frame"],
                         "stress_config_8192_tracks": klt_stress},
        }
        g16 = extra.get("frames_per_s_group_of_16", {}).get("tracker") if isinstance(extra.get("frames_per_s_group_of_16"), dict) else None
        out["roofline"]["frac_note"] = (
            "frac prices the single loop's own launches (about 2 000 tracks each), which last as long as their slowest "
            "track; the same kernel in launches that fill the GPU: %s of the LDS peak at 8 192 tracks (stress_config_8192_tracks), "
            "%s in a group of 16 loops (frames_per_s_group_of_16.tracker), where it issues at 0.60 of the vector issue rate "
            "(counter pass of profiles/r03b_group16)"
            % ("%.3f" % klt_stress["frac_lds"] if klt_stress else "n/a", "%.3f" % g16["frac_lds"] if g16 else "n/a"))
        if hyps_per_s is not None:
            out.update({
                "ransac_hypotheses_per_s": hyps_per_s,
                "ransac_config": {"workload": "RANSAC stress: %d P3P hypotheses x %d correspondences per step" % (h, n)},
                "roofline_ransac": {"bound": "valu_fp64",
                                    "bound_note": "compute-bound on the fp64 vector ALUs (no MFMA is issued: a 3x4 projection per "
                                                  "point is no matrix-core shape); 78.6 TFLOP/s is the fp64 vector peak, 72 TFLOP/s "
                                                  "what a register-only FMA loop sustains (tools/micro/fp64_peak.hip)",
                                    "kernel": "p3p_score_kernel<16, 1024> (RANSAC stress)", "achieved": achieved,
                                    "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_VALU_PEAK_TFLOPS,
                                    "traffic": traffic["bytes"] if traffic else None,
                                    "traffic_source": traffic["source"] if traffic else None,
                                    # 40 B per correspondence once, 4 x 12 doubles of K[R|t] in and 4 counts out per hypothesis
                                    "algorithmic_bytes": 40.0 * n + (384.0 + 16.0) * h,
                                    "kernel_ms": score_ms},
                "roofline_image_kernels": image_kernels,
                "bundle_adjustment": ba_runs})
        out.update(extra)
        if not args.no_cpu_baseline and world == 1 and not quick:
            X0, xy0 = init_points(vo, ctx, scene, frames_all[0], poses_all[0])
            cores = int(os.environ.get("OMP_NUM_THREADS", min(os.cpu_count() or 1, 16)))
            v, done, dt = cpu_baseline_frames(vo, scene, frames_all, poses_all, X0, xy0, threads=cores)
            out["cpu_baseline"] = {"value": v, "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": "%d frames of the same sequence through the CPU restatement of the loop "
                                             "(OpenMP in the tracker, scalar elsewhere), %.1f s" % (done, dt)}
            v1, done1, dt1 = cpu_baseline_frames(vo, scene, frames_all, poses_all, X0, xy0, budget_s=5.0, threads=1)
            out["cpu_baseline"]["single_thread"] = {"value": v1, "unit": "frames/s", "cores": 1,
                                                    "sample": "%d frames, %.1f s" % (done1, dt1)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
