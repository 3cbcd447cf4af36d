"""Batch evaluation over independent runs, the only part of the reference that shards across
GPUs: main_eval.m:22-23 loops `feats x repetition` full pipeline runs, each rebuilt from scratch
(main_eval.m:26-284) and saved on its own (main_eval.m:491-494).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on MI355X, "gloo" on
CPU).  Work items are dealt round-robin; there is no data-path collective.  The only exchange
is one all_gather of fixed-size result records at the end -- trajectory and per-frame counters,
the arrays the evaluation notebook reads from the saved runs.
"""
import numpy as np

T_MAX = 600           # frames a record can hold (the reference's longest run has 598)
HEADER = 16           # [item id, frames, seconds, fps, errs columns, total_time, frame_init, frame_ct, toc_param,
                      #  toc_bootstrap, toc_ct, fps_ct, failed, 3 spare] -- the scalars main_eval.m:491-494 saves; failed = 1
                      #  marks a run that ended with an error (main_eval(on_error="record"))
ERRS = 18             # errs 3 x 6 (quantitative_eval.m:27), column-major, unused columns NaN
RECORD_DOUBLES = HEADER + ERRS + 15 * T_MAX   # ... + est_trans 3xT + est_rot 9xT + num_X, num_C, num_new


TIMING_KEYS = ("total_time", "frame_init", "frame_ct", "toc_param", "toc_bootstrap", "toc_ct", "fps_ct")


def shard(items, rank, world, costs=None):
    """Static assignment of work items to ranks (a run's result never depends on it: every run carries its own
    seed).  Without `costs`: round-robin.  With `costs` (one number per item, e.g. its frame count): longest
    first -- the items in order of decreasing cost (ties: in order of their index), each to the rank with the
    smallest load so far (ties: the lowest rank) -- so that 20 runs of unequal length on 8 GPUs end together
    instead of 3 + 3 + 3 + 3 + 2 + 2 + 2 + 2 by count.  The share keeps the items' original order."""
    items = list(items)
    if costs is None:
        return [it for k, it in enumerate(items) if k % world == rank]
    if len(costs) != len(items):
        raise ValueError("one cost per item")
    load = [0.0] * world
    mine = []
    for k in sorted(range(len(items)), key=lambda k: (-float(costs[k]), k)):
        r = min(range(world), key=lambda r: (load[r], r))
        load[r] += float(costs[k])
        if r == rank:
            mine.append(k)
    return [items[k] for k in sorted(mine)]


def failed_record(item_id, seconds):
    """The record of a run that ended with an error (the reference's script would have stopped there,
    main_demo.m:389-391): no trajectory, failed = 1."""
    rec = pack_record(item_id, seconds, np.zeros((3, 0)), np.zeros((9, 0)), [], [], [])
    rec[3] = 0.0
    rec[12] = 1.0
    return rec


def pack_record(item_id, seconds, est_trans, est_rot, num_X, num_C, num_new, errs=None, timing=None):
    """One run -> a fixed-size float64 record.  errs: 3 x k (k <= 6) or None; timing: dict with any of
    total_time, frame_init, frame_ct, toc_param, toc_bootstrap, toc_ct, fps_ct."""
    T = est_trans.shape[1]
    if T > T_MAX:
        raise ValueError("run has %d frames, records hold %d" % (T, T_MAX))
    rec = np.zeros(RECORD_DOUBLES)
    rec[0], rec[1], rec[2] = item_id, T, seconds
    rec[3] = (T - 1) / seconds if seconds > 0 else 0.0
    e = rec[HEADER:HEADER + ERRS].reshape(6, 3)
    e[:] = np.nan
    if errs is not None:
        errs = np.asarray(errs, dtype=np.float64).reshape(3, -1)
        rec[4] = errs.shape[1]
        e[:errs.shape[1]] = errs.T
    for k, name in enumerate(TIMING_KEYS):
        rec[5 + k] = float((timing or {}).get(name, 0.0))
    body = rec[HEADER + ERRS:].reshape(15, T_MAX)
    body[0:3, :T] = est_trans
    body[3:12, :T] = est_rot
    for row, arr in ((12, num_X), (13, num_C), (14, num_new)):
        a = np.asarray(arr, dtype=np.float64)[:T]
        body[row, :len(a)] = a
    return rec


def unpack_record(rec):
    T = int(rec[1])
    body = rec[HEADER + ERRS:].reshape(15, T_MAX)
    out = dict(item=int(rec[0]), frames=T, seconds=float(rec[2]), fps=float(rec[3]), est_trans=body[0:3, :T].copy(),
               est_rot=body[3:12, :T].copy(), num_X=body[12, :T].copy(), num_C=body[13, :T].copy(),
               num_new=body[14, :T].copy(), errs=rec[HEADER:HEADER + ERRS].reshape(6, 3)[:int(rec[4])].T.copy())
    for k, name in enumerate(TIMING_KEYS):
        out[name] = float(rec[5 + k])
    out["failed"] = bool(rec[12])
    return out


def gather_records(local_records, n_items, device=None):
    """All ranks receive every record (list indexed by item id).  Ranks hold different numbers of
    items, so each contributes as many slots as the fullest rank holds (one MAX all_reduce of the
    counts), unused ones marked with id -1."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = [None] * n_items
        for r in local_records:
            out[int(r[0])] = unpack_record(r)
        return out
    world = dist.get_world_size()
    count = torch.tensor([len(local_records)], dtype=torch.int64)
    if device is not None:
        count = count.to(device)
    dist.all_reduce(count, op=dist.ReduceOp.MAX)
    slots = max(int(count.item()), 1)
    buf = np.zeros((slots, RECORD_DOUBLES))
    buf[:, 0] = -1.0
    for k, r in enumerate(local_records):
        buf[k] = r
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    out = [None] * n_items
    for p in parts:
        for r in p.cpu().numpy():
            if r[0] >= 0:
                out[int(r[0])] = unpack_record(r)
    return out


def run_items(mine, runner, concurrent=1):
    """[runner(item_id, item) for item_id, item in mine], up to `concurrent` of them at a time (one host
    thread each).  A single run leaves most of a GPU idle -- every stage of a frame is bound by latency --
    so independent runs overlap on it; the runner must give every thread a context of its own."""
    if concurrent <= 1 or len(mine) <= 1:
        return [runner(item_id, item) for item_id, item in mine]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=int(concurrent)) as pool:
        return list(pool.map(lambda p: runner(p[0], p[1]), mine))   # (in the order of `mine`)


def run_batch(items, runner, device=None, rank=None, world=None, concurrent=1, costs=None, share_runner=None, timing=None):
    """Runs `runner(item_id, item) -> record` for this rank's share of `items` and gathers all
    records on every rank.  `runner` builds and drives one full pipeline run.
    rank / world given explicitly (without a process group): only that share is run and returned, the
    other entries of the result are None -- one GPU can then rehearse every rank's share in turn.
    concurrent: runs a rank has in flight at once (see run_items).
    costs: see shard.  share_runner(mine) -> records (may be given instead of running item by item: the
    rank's whole share [(item_id, item), ...] at once, e.g. as groups of loops in lockstep).
    timing (a dict, filled in): seconds this rank spent running its share and in the gather."""
    import time
    import torch.distributed as dist
    run = (lambda mine: share_runner(mine)) if share_runner else (lambda mine: run_items(mine, runner, concurrent))
    if rank is not None:
        out = [None] * len(items)
        mine = shard(list(enumerate(items)), rank, world, costs)
        for (item_id, _), rec in zip(mine, run(mine)):
            out[item_id] = unpack_record(rec)
        return out
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    mine = shard(list(enumerate(items)), rank, world, costs)
    t0 = time.perf_counter()
    local = run(mine)
    t1 = time.perf_counter()
    out = gather_records(local, len(items), device=device)
    if timing is not None:
        timing.update(run_s=t1 - t0, gather_s=time.perf_counter() - t1, items=len(mine))
    return out
