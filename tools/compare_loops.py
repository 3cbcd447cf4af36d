"""Runs the stage-by-stage loop and the device-resident loop on the benchmark sequence with the same
seeds and prints where (if anywhere) they part, plus the final position errors."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
vo = importlib.import_module("visual_odometry_pipeline_amd")
ctx = vo.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 46
scene, frames, poses = bench.make_sequence(vo, n)
X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0])
hp = vo.default_hyper_paras("kitti")
for seed in range(4):
    a = vo.VOPipeline(vo.synth.K_KITTI, frames[0], X0, xy0, poses[0], hyper_paras=hp, rng=np.random.default_rng(seed), ctx=ctx)
    b = vo.VOPipelineDevice(vo.synth.K_KITTI, frames[0], X0, xy0, poses[0], hyper_paras=hp, rng=np.random.default_rng(seed), ctx=ctx)
    first = None
    for i in range(1, n):
        Ta, Tb = a.process_frame(frames[i]), b.process_frame(frames[i])
        same = (a.S.num_X[-1] == b.num_X[-1]) and (a.S.num_C[-1] == b.num_C[-1])
        d = np.abs(Ta - Tb).max()
        if first is None and (not same or d > 1e-6):
            first = (i, a.S.num_X[-1], b.num_X[-1], a.S.num_C[-1], b.num_C[-1], d)
    ea = np.linalg.norm(Ta[:3, 3] - poses[n - 1][:3, 3])
    eb = np.linalg.norm(Tb[:3, 3] - poses[n - 1][:3, 3])
    print("seed", seed, "first difference", first, "final errors %.3f %.3f" % (ea, eb), "landmarks", a.S.num_X[-1], b.num_X[-1], flush=True)
    a.close(); b.close()
