"""Throughput of a group of loops (VOGroupDevice) on the headline workload: S sequences of the KITTI-shape
rendered corridor advanced in lockstep, aggregate frames/s against the single loop's.
python tools/group_loop.py [--frames 60] [--seqs 1,2,4,8,16] [--distinct] [--split 2]
--split N: the S sequences as N groups of S/N on N contexts, driven by one host thread in turn (one group's
localisation chain and tail, which leave the GPU idle, beside another group's trackers)."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--seqs", default="1,2,4,8,16")
    ap.add_argument("--distinct", action="store_true", help="every sequence its own rendered corridor")
    ap.add_argument("--split", type=int, default=1)
    ap.add_argument("--probe", action="store_true", help="call vohip_klt_stats after every configuration (a measuring build prints its phase clocks there)")
    args = ap.parse_args()
    import torch
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    ctx = vo.Context(0)
    n = args.frames + args.warmup + 1
    worlds = {}

    def world(item):
        if item not in worlds:
            scene, frames, poses = bench.make_sequence(vo, n, item)
            d = [torch.from_numpy(f).cuda() for f in frames[:n]]
            X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0])
            worlds[item] = (d, X0, xy0, poses[0])
        return worlds[item]

    hp = vo.default_hyper_paras("kitti")
    hp.r_discard_redundant = bench.KITTI_EXCLUSION_RADIUS
    out = {}
    ctxs = [ctx] + [vo.Context(0) for _ in range(args.split - 1)]
    for S in [int(x) for x in args.seqs.split(",")]:
        N = args.split if S % args.split == 0 and S >= args.split else 1
        per = S // N
        gs = [vo.VOGroupDevice(vo.synth.K_KITTI, (vo.synth.KITTI_H, vo.synth.KITTI_W), per, n, hyper_paras=hp, ctx=ctxs[k])
              for k in range(N)]
        for rep in range(2):   # (the first run loads code objects and maps buffers)
            for s in range(S):
                d, X0, xy0, T0 = world(s if args.distinct else 0)
                gs[s // per].add_sequence(s % per, d, X0, xy0, T0, rng=np.random.default_rng(1000 + s))
            for _ in range(args.warmup):
                for g in gs:
                    g.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0, w0 = sum(g.host_enqueue_us for g in gs), sum(g.host_wait_us for g in gs)
            pairs = 0
            for g in gs:
                g.begin_step()
            for k in range(args.frames):
                for g in gs:
                    pairs += len(g.finish_step())
                    if k + 1 < args.frames:
                        g.begin_step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[S] = dict(frames_per_s=pairs / dt, ms_per_step=1e3 * dt / args.frames, groups=N,
                      host_enqueue_us_per_step=(sum(g.host_enqueue_us for g in gs) - e0) / args.frames,
                      host_wait_us_per_step=(sum(g.host_wait_us for g in gs) - w0) / args.frames,
                      landmarks=int(np.mean([q.num_X[-1] for g in gs for q in g.seqs])),
                      candidates=int(np.mean([q.num_C[-1] for g in gs for q in g.seqs])))
        print(S, json.dumps(out[S]), flush=True)
        if args.probe:
            import ctypes as C
            tr, it = C.c_longlong(0), C.c_longlong(0)
            vo._lib.load().vohip_klt_stats(ctx.handle, C.byref(tr), C.byref(it), 1)
        for g in gs:
            g.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
