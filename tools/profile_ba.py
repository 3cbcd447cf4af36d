"""Runs the sliding-window bundle adjustment a few times (for rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
vo = importlib.import_module("visual_odometry_pipeline_amd")
ctx = vo.Context(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 300
s0, obs, _ = vo.synth.ba_problem(F=5, m=m, seed=9)
for _ in range(10):
    out, info = vo.runBA(s0, obs, vo.synth.K_KITTI, return_stats=True, ctx=ctx)
print(info)
