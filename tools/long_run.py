"""A reference-length run (main_eval.m:370-462 processes KITTI 05 frames 3..540) at the benchmark's headline
settings (1241x376, first-frame thinning cell 10 px, exclusion radius 5.5 px): the device-resident loop against
the CPU restatement of the loop, frame by frame.  Prints the first frame at which counts / ids / poses differ.
python tools/long_run.py [frames]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 540
    resync = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # hand the device's state to the restatement every so many frames
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    from oracle import pipeline_oracle as po
    from oracle import vo_oracle
    vo_oracle.build()
    ctx = vo.Context(0)
    t0 = time.time()
    scene, frames, poses = bench.make_sequence(vo, n + 1, 0)
    print("rendered %d frames in %.1f s" % (len(frames), time.time() - t0), flush=True)
    hp = bench.kitti_hyper_paras(vo)
    X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0])
    ref = po.OraclePipeline(vo.synth.K_KITTI, frames[0], X0, xy0, poses[0], hp, 3, 2, np.random.default_rng(100), vo.draw_triples,
                            vo.draw_octets)
    dev = vo.VOPipelineDevice(vo.synth.K_KITTI, frames[0], X0, xy0, poses[0], hyper_paras=hp, rng=np.random.default_rng(100), ctx=ctx)
    worst_t = worst_r = 0.0
    t0 = time.time()
    for i in range(1, n):
        try:
            Tr = ref.process_frame(frames[i])
        except RuntimeError as e:
            print("frame %d: the restatement stops: %s" % (i, e), flush=True)
            Td = dev.process_frame(frames[i], next_image=frames[i + 1])
            print("  the device loop: %d tracked, %d inliers, %d iterations, %d landmarks" % (
                dev.last_pose.n_tracked, dev.last_pose.n_inliers, dev.last_pose.n_iter, dev.num_X[-1]), flush=True)
            break
        Td = dev.process_frame(frames[i], next_image=frames[i + 1])
        same = dev.num_X[-1] == ref.S.num_X[-1] and dev.num_C[-1] == ref.S.num_C[-1]
        dt = float(np.linalg.norm(Tr[:3, 3] - Td[:3, 3]))
        dR = Tr[:3, :3].T @ Td[:3, :3]
        dr = float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
        if not same or dt > 1e-5 or dr > 1e-5:
            print("frame %d differs: landmarks %d / %d, candidates %d / %d, |dt| %.3g, angle %.3g, n_iter %d" % (
                i, dev.num_X[-1], ref.S.num_X[-1], dev.num_C[-1], ref.S.num_C[-1], dt, dr, dev.last_pose.n_iter), flush=True)
            break
        if resync and i % resync == 0:
            S = dev.S
            ref.S.P, ref.S.X, ref.S.C, ref.S.F, ref.S.T = S.P.copy(), S.X.copy(), S.C.copy(), S.F.copy(), S.T.copy()
            ref.L_pts = np.fliplr(ref.S.P.T)
        if i % 50 == 0:
            print("frame %d: %d landmarks, %d candidates, worst |dt| %.3g, worst angle %.3g, %.1f s" % (
                i, dev.num_X[-1], dev.num_C[-1], worst_t, worst_r, time.time() - t0), flush=True)
    else:
        ids_equal = np.array_equal(dev.S.X[3], ref.S.X[3])
        print("all %d frames agree: ids equal %s, worst |dt| %.3g, worst angle %.3g, position error %.2f m" % (
            n - 1, ids_equal, worst_t, worst_r, np.linalg.norm(dev.est_trans[-1] - poses[n - 1][:3, 3])))
    dev.close()


if __name__ == "__main__":
    main()
