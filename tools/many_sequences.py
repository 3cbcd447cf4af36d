"""Aggregate frames/s of n independent sequences on one GPU (one host thread and context each).
usage: [GPU_MAX_HW_QUEUES=q] python3 tools/many_sequences.py n_seq [n_seq ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
vo = importlib.import_module("visual_odometry_pipeline_amd")
torch.cuda.set_device(0)
scene, frames, poses = bench.make_sequence(vo, 48)
for n in [int(a) for a in sys.argv[1:]]:
    fps = bench.bench_frames_concurrent(vo, torch, scene, frames, poses, 40, 5, n)
    print("queues %s sequences %d: %.0f frames/s" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), n, fps), flush=True)
