"""Corner detection alone, 200 calls on one KITTI-shape frame (run under rocprofv3 --kernel-trace --stats
for the per-kernel times, or with --pmc FETCH_SIZE / WRITE_SIZE for the traffic)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
ctx = vo.Context(0)
scene, frames, poses = bench.make_sequence(vo, 8)
out = bench.bench_image_kernels(vo, torch, ctx, torch.from_numpy(frames[0]).cuda(), reps=200)
print(out["corner_detection"], out["fast_detection"], out["pyramid_build"])
