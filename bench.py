#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM.
Prints ONE JSON line on rank 0.  The CPU oracle is used only for the cpu_baseline leg.
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
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X vector fp64, 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def ransac_flops(n, h):
    """Algorithmic work of hypothesis generation + scoring: 29 flop per (hypothesis, solution,
    point) in fused form x 4 solutions + ~1600 flop for P3P and the quartic."""
    return (116.0 * n + 1600.0) * h


def bench_ransac(vo, torch, ctx, n, h, steps, warmup, seed):
    """Stress configuration: h P3P hypotheses scored against n correspondences per step."""
    lib = vo._lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    kp_rc, X, _, _, _ = vo.synth.pnp_problem(n, seed=seed)
    samples = vo.synth.sample_triples(n, h, seed=42 + seed)
    d_kp = torch.from_numpy(np.ascontiguousarray(kp_rc.T)).to(dev)
    d_X = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    d_s = torch.from_numpy(samples).to(dev)
    d_c = torch.zeros((h, 4), dtype=torch.int32, device=dev)
    Kc = vo._lib.colmajor(vo.synth.K_KITTI)
    Kp = vo._lib.dptr(Kc)

    def step():
        ctx.check(lib.vohip_ransac_p3p_score_dev(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), n, Kp,
                                                  d_s.data_ptr(), h, d_c.data_ptr()))

    def count_only():
        ctx.check(lib.vohip_ransac_p3p_count_dev(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), n, h,
                                                  d_c.data_ptr()))

    return step, count_only, (kp_rc, X, samples, d_c)


def cpu_baseline_ransac(kp_rc, X, K, samples, budget_s=12.0):
    """Times the CPU oracle (single thread, scalar C port) on a bounded sample of the same
    workload: as many of the hypotheses as fit in about `budget_s` seconds."""
    from oracle import vo_oracle
    vo_oracle.build()
    kp_uv = kp_rc[::-1].T.copy()
    Xc = np.ascontiguousarray(X.T)
    done = 0
    t0 = time.perf_counter()
    while done < len(samples):
        vo_oracle.ransac_hypothesis(kp_uv, Xc, K, samples[done])
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done / dt, done, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=8192)
    ap.add_argument("--hypotheses", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    stream = torch.cuda.current_stream().cuda_stream
    ctx = vo.Context(local_rank, stream=stream)

    n, h = args.points, args.hypotheses
    step, count_only, (kp_rc, X, samples, d_c) = bench_ransac(vo, torch, ctx, n, h, args.steps, args.warmup,
                                                            seed=100 + rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel (inlier scoring) alone, HIP events on the stream it is launched on
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    reps = max(args.steps, 20)
    count_only()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        count_only()
    ev1.record()
    torch.cuda.synchronize()
    score_ms = ev0.elapsed_time(ev1) / reps
    achieved = 116.0 * n * h / (score_ms * 1e-3) / 1e12

    if rank == 0:
        hyps_per_s = world * h * args.steps / elapsed
        out = {
            "metric": METRIC,
            "value": hyps_per_s,
            "unit": "RANSAC hypotheses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "RANSAC stress: %d P3P hypotheses x %d correspondences per step, "
                                   "70%% inliers, KITTI camera" % (h, n),
                       "points": n, "hypotheses_per_step": h},
            "roofline": {"bound": "valu_fp64", "kernel": "p3p_score_kernel", "achieved": achieved,
                         "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VALU_PEAK_TFLOPS, "traffic": None,
                         "kernel_ms": score_ms},
        }
        if not args.no_cpu_baseline and world == 1:
            v, done, dt = cpu_baseline_ransac(kp_rc, X, vo.synth.K_KITTI, samples)
            out["cpu_baseline"] = {"value": v, "unit": "RANSAC hypotheses/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the %d hypotheses of one step (%d points each), %.1f s"
                                             % (done, h, n, dt)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
