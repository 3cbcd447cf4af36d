"""The image stages alone (bench.bench_image_kernels: single-frame launch sequences, then 256 frames per launch
sequence), for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/batch_image_kernels.py"""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ctx = vo.Context(0, stream=stream.cuda_stream)
    scene, frames, poses = bench.make_sequence(vo, 47)
    d = [torch.from_numpy(f).cuda() for f in frames[:32]]
    out = bench.bench_image_kernels(vo, torch, ctx, d[0], d_frames=d)
    print(json.dumps(out["batched_256_frames"], indent=1))
