"""A long KITTI-shape sequence with bundle adjustment (main_BA.m's window) through the device-resident loop
(window kept by the library) and through the stage-by-stage loop (host mirror of VO_bundle_adjust.m): counts,
window sizes and trajectories side by side.  Usage: python tools/long_ba_loop.py [frames]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    ctx = vo.Context(0)
    scene, frames, poses = bench.make_sequence(vo, n + 2)
    X0, xy = bench.init_points(vo, ctx, scene, frames[0], poses[0])
    hp = vo.default_hyper_paras("kitti", is_BA=True)
    dev = vo.VOPipelineDevice(scene.K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(42), ctx=ctx,
                              window_size=5, keyframe_d=1)
    stg = vo.VOPipeline(scene.K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(42), ctx=ctx,
                        window_size=5, keyframe_d=1)
    worst = 0.0
    for i in range(1, n + 1):
        Td = dev.process_frame(frames[i], next_image=frames[i + 1])
        Ts = stg.process_frame(frames[i])
        same = dev.num_X[-1] == stg.S.num_X[-1] and dev.num_C[-1] == stg.S.num_C[-1]
        worst = max(worst, float(np.abs(Td - Ts).max()))
        if not same or i % 20 == 0:
            print("frame %3d  landmarks %d / %d  candidates %d / %d  |T_dev - T_stage| %.2e  position error %.3f m" %
                  (i, dev.num_X[-1], stg.S.num_X[-1], dev.num_C[-1], stg.S.num_C[-1], np.abs(Td - Ts).max(),
                   np.linalg.norm(Td[:3, 3] - poses[i][:3, 3])), flush=True)
        if not same:
            print("the loops took different decisions at frame", i)
            break
    et_d, _ = dev.trajectory()
    et_s, _ = stg.trajectory()
    k = min(et_d.shape[1], et_s.shape[1])
    print("BA runs %d, window: %d keyframes, %d / %d landmarks; largest pose difference %.2e, trajectories differ by at most %.2e" %
          (dev.ba_runs, dev.B.num_key, dev.B.m, stg.B.m, worst, np.abs(et_d[:, :k] - et_s[:, :k]).max()))


if __name__ == "__main__":
    main()
