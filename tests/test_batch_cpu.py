"""The multi-process (N > 1) path on CPU: two gloo ranks shard a run grid and gather the records."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, importlib
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
dist.init_process_group("gloo")
batch = importlib.import_module("visual_odometry_pipeline_amd").batch
items = ["run%%d" %% k for k in range(5)]
def runner(item_id, item):
    T = 3 + item_id
    et = np.arange(3 * T, dtype=np.float64).reshape(3, T) + 100 * item_id
    er = np.tile(np.eye(3).reshape(9, 1), (1, T))
    return batch.pack_record(item_id, 0.5 * (item_id + 1), et, er, [10] * T, [20] * T, [1] * T)
recs = batch.run_batch(items, runner)
assert len(recs) == 5 and all(r is not None for r in recs)
for k, r in enumerate(recs):
    assert r["item"] == k and r["frames"] == 3 + k and r["est_trans"][0, 0] == 100 * k
    assert abs(r["fps"] - (2 + k) / (0.5 * (k + 1))) < 1e-12 and r["num_C"][0] == 20
mine = batch.shard(list(range(5)), dist.get_rank(), dist.get_world_size())
assert mine == ([0, 2, 4] if dist.get_rank() == 0 else [1, 3])
# longest first: one long run and six short ones -- rank 0 holds 1 item, rank 1 holds 6 (more than
# ceil(7 / 2) slots), records of different lengths, one run failed, the share run as a whole
items = ["long"] + ["short%%d" %% k for k in range(6)]
costs = [60, 10, 10, 10, 10, 10, 10]
def runner2(item_id, item):
    if item_id == 4:
        return batch.failed_record(item_id, 0.25)
    T = costs[item_id] // 2
    et = np.full((3, T), float(item_id)); er = np.tile(np.eye(3).reshape(9, 1), (1, T))
    return batch.pack_record(item_id, 1.0, et, er, [item_id] * T, [2] * T, [3] * T, errs=np.full((3, item_id %% 4), 0.5))
seen = []
def share_runner(mine):
    seen.extend(i for i, _ in mine)
    return [runner2(i, it) for i, it in mine]
timing = {}
recs = batch.run_batch(items, None, costs=costs, share_runner=share_runner, timing=timing)
assert seen == ([0] if dist.get_rank() == 0 else [1, 2, 3, 4, 5, 6]), seen
assert timing["items"] == len(seen) and timing["gather_s"] >= 0.0
assert len(recs) == 7 and all(r is not None for r in recs)
for k, r in enumerate(recs):
    assert r["item"] == k and r["failed"] == (k == 4)
    if k != 4:
        assert r["frames"] == costs[k] // 2 and r["num_X"][0] == k and r["errs"].shape == (3, k %% 4)
    else:
        assert r["frames"] == 0 and r["est_trans"].shape == (3, 0)
dist.barrier()
dist.destroy_process_group()
print("rank ok")
'''


def test_two_rank_gloo_batch(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("rank ok") == 2


def test_record_roundtrip_and_single_process(vo):
    b = vo.batch
    rng = np.random.default_rng(0)
    et, er = rng.normal(size=(3, 7)), rng.normal(size=(9, 7))
    r = b.unpack_record(b.pack_record(3, 2.0, et, er, range(7), range(7), range(7)))
    np.testing.assert_array_equal(r["est_trans"], et)
    np.testing.assert_array_equal(r["est_rot"], er)
    assert r["fps"] == 3.0 and r["item"] == 3 and r["errs"].shape == (3, 0) and r["fps_ct"] == 0.0
    errs = rng.normal(size=(3, 4))
    r = b.unpack_record(b.pack_record(3, 2.0, et, er, range(7), range(7), range(7), errs=errs,
                                      timing=dict(fps_ct=12.5, toc_ct=0.5, frame_ct=6, total_time=0.75)))
    np.testing.assert_array_equal(r["errs"], errs)
    assert (r["fps_ct"], r["toc_ct"], r["frame_ct"], r["total_time"], r["toc_param"]) == (12.5, 0.5, 6.0, 0.75, 0.0)
    recs = b.run_batch(["a", "b"], lambda i, it: b.pack_record(i, 1.0, et, er, range(7), range(7), range(7)))
    assert [x["item"] for x in recs] == [0, 1]
    assert b.shard(list(range(7)), 1, 3) == [1, 4]
    # longest first: every item lands on exactly one rank whatever the number of ranks, shares keep item order,
    # and 20 runs of unequal length on 8 ranks end within one short run of each other
    costs = [540] * 4 + [270] * 8 + [135] * 8
    for world in (1, 2, 3, 8):
        shares = [b.shard(list(range(20)), r, world, costs) for r in range(world)]
        assert sorted(sum(shares, [])) == list(range(20)) and all(s == sorted(s) for s in shares)
        loads = [sum(costs[k] for k in s) for s in shares]
        assert max(loads) - min(loads) <= 270, (world, loads)
    assert max(sum(costs[k] for k in b.shard(list(range(20)), r, 8, costs)) for r in range(8)) == 675
    assert max(sum(costs[k] for k in b.shard(list(range(20)), r, 8)) for r in range(8)) == 945   # (round-robin)
