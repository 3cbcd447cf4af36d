"""k independent KITTI-shape frame loops on ONE GPU, one PROCESS each (bench.py's
`frames_per_s_4_sequences_at_once` runs them as threads of one process, which share the runtime's launch path).
The parent never touches the GPU; every child builds its loop, waits for a common wall-clock start, runs
`steps` frames and prints its start and end.  Aggregate = k * steps / (last end - start).
    python tools/processes_at_once.py [k=4] [steps=200]"""
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(idx, steps, start):
    import numpy as np
    import torch
    import bench
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    warmup = 5
    scene, frames, poses = bench.make_sequence(vo, warmup + steps + 2)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = vo.Context(0, stream=stream.cuda_stream)
        d_frames = [torch.from_numpy(f).cuda() for f in frames]
        X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0])
        hp = vo.default_hyper_paras("kitti", is_BA=False)
        hp.r_discard_redundant = bench.KITTI_EXCLUSION_RADIUS
        pipe = vo.VOPipelineDevice(vo.synth.K_KITTI, d_frames[0], X0, xy0, poses[0], hyper_paras=hp,
                                   rng=np.random.default_rng(1000 + idx), ctx=ctx)
        for i in range(1, warmup + 1):
            pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
        torch.cuda.synchronize()
        late = time.time() > start
        while time.time() < start:
            time.sleep(0.0005)
        t0 = time.time()
        for i in range(warmup + 1, warmup + steps + 1):
            pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
        t1 = time.time()
        pipe.close()
        ctx.close()
    print(json.dumps(dict(idx=idx, t0=t0, t1=t1, late=late, frames_per_s=steps / (t1 - t0))), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]))
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    import bench
    bench.make_sequence(importlib.import_module("visual_odometry_pipeline_amd"), 5 + steps + 2)   # fills the frame cache (CPU only)
    start = time.time() + 45.0    # the children import torch and build their loops before this
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(i), str(steps), repr(start)],
                              stdout=subprocess.PIPE, text=True) for i in range(k)]
    rows = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        if p.returncode != 0:
            raise SystemExit("child failed: %d" % p.returncode)
        rows.append(json.loads(out.strip().splitlines()[-1]))
    end = max(r["t1"] for r in rows)
    print(json.dumps(dict(processes=k, steps=steps, any_late=any(r["late"] for r in rows),
                          each=[round(r["frames_per_s"]) for r in rows],
                          aggregate_frames_per_s=k * steps / (end - min(r["t0"] for r in rows)))))


if __name__ == "__main__":
    main()
