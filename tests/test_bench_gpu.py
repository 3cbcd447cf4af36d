"""bench.py's multi-process path rehearsed on one GPU: two ranks (gloo for the gather, both on GPU 0) each
run their own work item of the batch grid and rank 0 prints the JSON line the driver parses."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_rehearsal(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", VOHIP_BENCH_BACKEND="gloo", VOHIP_CACHE=str(tmp_path / "cache"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["unit"] == "frames/s" and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["vs_baseline"] is None
    assert rec["roofline"]["kernel"].startswith("klt_track_kernel") and 0 < rec["roofline"]["frac"] < 1
    assert rec["roofline_ransac"]["bound"] == "valu_fp64"
    # (two tracker launches per frame -- the kept landmarks', and the appended landmarks' + the candidates' -- plus the first
    # frame's own, plus any that a frame in need of more hypotheses repeats)
    assert 2 * (4 + 2) <= rec["roofline"]["launches"] <= 2 * (4 + 2) + 4 and rec["roofline"]["tracks_per_launch"] > 100
    grid = rec["run_grid_groups_of_4"]
    assert grid["items_per_rank"] == 4 and grid["frames_per_s"] > 0
    assert rec["config"]["gather_ms_inside_the_timed_region"] >= 0
