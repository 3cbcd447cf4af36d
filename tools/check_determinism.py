"""Runs the device-resident loop over a long KITTI-shape sequence three times -- twice with the look-ahead
(following frame prepared, its trackers enqueued behind the tail), once with the plain call -- and checks
that trajectories, counts and final states agree bit for bit.  Usage: python tools/check_determinism.py [frames]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (sequence cache and point initialisation of the benchmark)


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    ctx = vo.Context(0)
    scene, frames, poses = bench.make_sequence(vo, n + 2)
    X0, xy = bench.init_points(vo, ctx, scene, frames[0], poses[0])
    d_frames = [torch.from_numpy(f).cuda() for f in frames]
    runs = []
    for mode in ("ahead", "ahead", "plain"):
        pipe = vo.VOPipelineDevice(scene.K, d_frames[0], X0, xy, poses[0], hyper_paras=vo.default_hyper_paras("kitti"),
                                   rng=np.random.default_rng(42), ctx=ctx)
        for i in range(1, n + 1):
            pipe.process_frame(d_frames[i], next_image=d_frames[i + 1] if mode == "ahead" else None)
        S = pipe.S
        runs.append((mode, S, list(pipe.num_X), list(pipe.num_C)))
        pipe.close()
    ref = runs[0]
    ok = True
    for mode, S, nx, nc in runs[1:]:
        same = nx == ref[2] and nc == ref[3]
        for k in ("X", "P", "C", "F", "T", "est_trans", "est_rot"):
            same = same and np.array_equal(getattr(S, k), getattr(ref[1], k))
        print("%s vs ahead: %s" % (mode, "identical" if same else "DIFFERENT"))
        ok = ok and same
    print("frames %d, landmarks %d, candidates %d, end position error %.3f m" %
          (n, ref[2][-1], ref[3][-1], np.linalg.norm(ref[1].est_trans[:, -1] - poses[n][:3, 3])))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
