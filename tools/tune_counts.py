"""Steady-state landmark / candidate counts of the frame loop for a few detector and exclusion-radius
settings (used to hold the benchmark workloads at their nominal sizes).
Usage: python tools/tune_counts.py [stress]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    ctx = vo.Context(0)
    scene, frames, poses = bench.make_sequence(vo, 47)
    d_frames = [torch.from_numpy(f).cuda() for f in frames]
    stress = len(sys.argv) > 1 and sys.argv[1] == "stress"
    grid = ([(2e-5, 5.0, None), (1e-5, 4.0, None), (5e-6, 3.0, None), (2e-6, 3.0, None), (1e-6, 2.5, None), (1e-6, 2.0, None)] if stress else
            [(2e-4, 10.0, 14.0), (2e-4, 8.0, 12.0), (2e-4, 7.0, 12.0), (2e-4, 6.0, 10.0), (2e-4, 5.0, 10.0)])
    for mq, r, cell in grid:
        hp = vo.default_hyper_paras("kitti")
        hp.feature_extract_options = ["MinQuality", mq]
        hp.r_discard_redundant = r
        _, kp = vo.genKeypoints(frames[0], "Harris", ["MinQuality", mq], ctx=ctx)
        xy = kp.Location.astype(np.float64)
        if cell:
            xy = xy[bench.thin_by_grid(xy, kp.Metric, cell)]
        elif len(xy) > 16000:
            xy = xy[np.sort(np.argsort(-kp.Metric, kind="stable")[:16000])]
        X0 = scene.backproject(poses[0], xy)
        try:
            pipe = vo.VOPipelineDevice(vo.synth.K_KITTI, d_frames[0], X0, xy, poses[0], hyper_paras=hp,
                                       rng=np.random.default_rng(1), ctx=ctx, cap_landmarks=16384, cap_candidates=16384)
            for i in range(1, 45):
                pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
            print("min_quality %g r %g cell %s: detections %d start %d -> landmarks %.0f candidates %.0f (last 20 frames) err %.3f m"
                  % (mq, r, cell, len(kp.Location), len(xy), np.mean(pipe.num_X[-20:]), np.mean(pipe.num_C[-20:]),
                     np.linalg.norm(pipe.est_trans[-1] - poses[44][:3, 3])), flush=True)
            pipe.close()
        except Exception as e:
            print("min_quality %g r %g cell %s: %s" % (mq, r, cell, e), flush=True)


if __name__ == "__main__":
    main()
