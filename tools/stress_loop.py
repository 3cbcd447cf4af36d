"""Runs only the frame loop at the stress sizes of the benchmark (dense corners, 4096 hypotheses per frame);
meant to be run under rocprofv3 --kernel-trace --stats.  Usage: python tools/stress_loop.py [steps]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    ctx = vo.Context(0)
    scene, frames, poses = bench.make_sequence(vo, steps + 7)
    el, info = bench.bench_frames(vo, torch, ctx, scene, frames, poses, steps, 5, 100, torch.cuda.synchronize,
                                  lambda pipe, seconds: [], stress_hypotheses=4096)
    print("stress loop: %.0f frames/s, %s" % (steps / el, info))


if __name__ == "__main__":
    main()
