"""Where the host time of a grouped batch evaluation goes: main_eval(group=8) over the bench's KITTI-shape
sequence under cProfile (cumulative times of the stages of run_group)."""
import cProfile
import importlib
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 212
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    ctx = vo.Context(0)
    scene, frames, poses = bench.make_sequence(vo, n, 0)
    seq = dict(name="kitti", images=frames, K=vo.synth.K_KITTI,
               ground_truth=np.array([[T[0, 3], T[2, 3]] for T in poses[:len(frames)]]))
    feats = ("Harris", "MinEigen", "FAST")
    vo.main_eval([dict(seq, images=frames[:12])], feats=feats, repetition=1, seed=10, ctx=ctx, group=8)
    for label, kw in (("one at a time", {}), ("groups of 8", dict(group=8)), ("groups of 12", dict(group=12))):
        t0 = time.perf_counter()
        recs = vo.main_eval([seq], feats=feats, repetition=4, seed=10, ctx=ctx, **kw)
        dt = time.perf_counter() - t0
        print("%-14s %.3f s  %.1f runs/s  %.0f frames/s" % (label, dt, len(recs) / dt, sum(r["frame_ct"] for r in recs) / dt))
    pr = cProfile.Profile()
    pr.enable()
    vo.main_eval([seq], feats=feats, repetition=4, seed=10, ctx=ctx, group=8)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
