"""The parking configuration's loop alone (bench.bench_parking), without and with bundle adjustment."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
ctx = vo.Context(0)
for ba in (False, True):
    r = bench.bench_parking(vo, torch, ctx, 40, 5, ba)
    print("BA" if ba else "no BA", "%.0f frames/s" % r["frames_per_s"], r["tracked_landmarks"], r["candidates"])
