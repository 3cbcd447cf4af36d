"""cProfile of one full run of the batch evaluation (run_sequence: bootstrap, loop, evaluation): where a run's
host time goes besides the frames.  Usage: python tools/profile_run_sequence.py"""
import cProfile
import importlib
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
ctx = vo.Context(0)
scene, frames, poses = bench.make_sequence(vo, 47)
gt = np.array([[T[0, 3], T[2, 3]] for T in poses])
hp = vo.default_hyper_paras("kitti")
for k in range(2):
    vo.run_sequence(frames, vo.synth.K_KITTI, hp, ground_truth=gt, rng=np.random.default_rng([10, k]), ctx=ctx)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for k in range(5):
    out = vo.run_sequence(frames, vo.synth.K_KITTI, hp, ground_truth=gt, rng=np.random.default_rng([10, k]), ctx=ctx)
pr.disable()
print("ms per run %.2f (loop %.2f, bootstrap %.2f)" % (1e3 * (time.perf_counter() - t0) / 5, 1e3 * out["toc_ct"], 1e3 * out["toc_bootstrap"]))
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
