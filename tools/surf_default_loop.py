"""The frame loop with the scripts' own default feature setting (main_demo.m:25-26: SURF, MetricThreshold 10) on the
KITTI-shape rendered sequence: detections per frame, tracked points and frames/s.
    python3 tools/surf_default_loop.py [--frames 60] [--threshold 10]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--threshold", type=float, default=10.0)
    args = ap.parse_args()
    import torch
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    ctx = vo.Context(0)
    scene, frames, poses = bench.make_sequence(vo, args.frames + 12)
    feature = ("SURF", ["MetricThreshold", args.threshold])
    _, kp = vo.genKeypoints(frames[0], *feature, ctx=ctx)
    dev = torch.device("cuda", 0)
    d_frames = [torch.from_numpy(f).to(dev) for f in frames]
    hp = vo.default_hyper_paras("kitti")
    hp.feature_extract, hp.feature_extract_options = feature[0], list(feature[1])
    X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0], feature=feature)
    pipe = vo.VOPipelineDevice(vo.synth.K_KITTI, d_frames[0], X0, xy0, poses[0], hyper_paras=hp,
                               rng=np.random.default_rng(1), ctx=ctx, cap_landmarks=16384, cap_candidates=16384)
    for i in range(1, 11):
        pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
    t0 = time.perf_counter()
    for i in range(11, 11 + args.frames):
        pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
    dt = time.perf_counter() - t0
    out = {"metric_threshold": args.threshold, "detections_in_frame_0": int(kp.Count), "frames": args.frames,
           "frames_per_s": args.frames / dt, "landmarks": int(np.mean(pipe.num_X[-args.frames:])),
           "candidates": int(np.mean(pipe.num_C[-args.frames:])),
           "final_position_error_m": float(np.linalg.norm(pipe.est_trans[-1] - poses[10 + args.frames][:3, 3]))}
    pipe.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
