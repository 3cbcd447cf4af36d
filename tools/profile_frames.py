"""cProfile of the frame loop (host-side cost breakdown)."""
import cProfile, importlib, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
vo = importlib.import_module("visual_odometry_pipeline_amd")
ctx = vo.Context(0)
scene, frames, poses = bench.make_sequence(vo, 46)
d_frames = [torch.from_numpy(f).cuda() for f in frames]
X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0])
pipe = vo.VOPipeline(vo.synth.K_KITTI, d_frames[0], X0, xy0, poses[0], hyper_paras=vo.default_hyper_paras("kitti"), rng=np.random.default_rng(1), ctx=ctx)
for i in range(1, 6):
    pipe.process_frame(d_frames[i])
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(6, 46):
    pipe.process_frame(d_frames[i])
pr.disable()
print("ms/frame", 1e3 * (time.perf_counter() - t0) / 40)
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
