"""The KITTI-shape frame loop of the benchmark with the stored sequence prepared ahead (VOPipelineDevice.prepare:
all pyramids and corner lists from batched launches), preparation timed with the loop, against the plain
look-ahead loop.  Usage: python tools/prepared_loop.py [n_sequences_at_once]"""
import importlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps, warmup = 200, 5
scene, frames, poses = bench.make_sequence(vo, warmup + steps + 2)
d_frames = [torch.from_numpy(f).cuda() for f in frames]
for prepared in (False, True, False, True):
    seconds = [0.0] * n_seq
    gate = threading.Barrier(n_seq)

    def worker(idx):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            ctx = vo.Context(0, stream=stream.cuda_stream)
            X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0])
            hp = vo.default_hyper_paras("kitti")
            hp.r_discard_redundant = bench.KITTI_EXCLUSION_RADIUS
            pipe = vo.VOPipelineDevice(vo.synth.K_KITTI, d_frames[0], X0, xy0, poses[0], hyper_paras=hp,
                                       rng=np.random.default_rng(100 + idx), ctx=ctx)
            for i in range(1, warmup + 1):
                pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
            gate.wait()
            t0 = time.perf_counter()
            if prepared:
                pipe.prepare(d_frames[warmup + 1:])
            for i in range(warmup + 1, warmup + steps + 1):
                pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
            seconds[idx] = time.perf_counter() - t0
            gate.wait()
            pipe.close()
            ctx.close()

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(n_seq)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print("%d sequence(s), %s: %.0f frames/s (preparation inside the timed region)" %
          (n_seq, "prepared" if prepared else "frame by frame", n_seq * steps / max(seconds)), flush=True)
