#!/usr/bin/env python3
"""First contact with a multi-GPU node, made diagnosable: every collective shape the sharded fit and bench.py use, on the
process groups they would create, checked value by value -- before anything of the workload is built.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29531 \
        tools/rccl_selftest.py [--workloads cfg3,cfg5] [--rounds 20]

(one rank per GPU, backend nccl = RCCL over xGMI; PHMRF_DIST_BACKEND=gloo runs the same checks on host tensors, which is
how tests/test_host_logic.py exercises this file on a CPU-only box).  What it does, in the order a failure would otherwise
surface in the middle of bench.py:

  1. process group FIRST (before any HIP call of this process), then the device of LOCAL_RANK;
  2. the world: the E-step's statistics message (K (1 + S + S S) + 5 float64 = 3.4 KB at K = 20, S = 4; dist.Reducer), the
     M-step's rows (K (3B + 2 + S + S S) + 1 float64; phyloHMRF._do_mstep), the label gather (one byte per node;
     Reducer.allreduce_bytes), a max-reduce and a barrier (bench.py's timing);
  3. the TILE SUB-GROUPS: for each workload, the groups tiles.plan / tiles.assign give at this world size -- created by every
     rank in block order exactly as bench.py and phyloHMRF do (dist.new_group is collective) -- and on each the int64 payload of
     a lockstep round (per tile 128 counters + energy + two boundary rows: 11 KB at 50 kb; tiles.TileGroup.finish_round);
  4. each message is all-reduced --rounds times; rank 0 prints min / median / max microseconds per call and every rank checks
     the sums against the closed form.  A mismatch, an exception or a rank that does not arrive names the step it was in.

Exit code 0 and one JSON line from rank 0 = the transport is fine; anything else = read the last "[selftest rank r] step"
lines of the ranks' stderr."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="cfg3,cfg5")
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--split-above", type=float, default=1.0)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PHMRF_DIST_BACKEND", "nccl")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL fails with hipIpcGetMemHandle otherwise
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    step = ["start"]

    def say(msg):
        step[0] = msg
        sys.stderr.write("[selftest rank %d] %s\n" % (rank, msg))
        sys.stderr.flush()

    import torch
    import torch.distributed as dist
    say("init_process_group(%s), world %d -- before any HIP call" % (backend, world))
    if backend == "nccl":
        if os.environ.get("PHMRF_ONE_GPU") == "1":
            local_rank = 0
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        torch.cuda.set_device(dev)
        coll_dev = dev
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
        coll_dev = torch.device("cpu")
    report = {"backend": backend, "world": world, "device": str(coll_dev), "messages": {}, "tile_groups": {}}

    def timed_allreduce(name, make, expect, group=None, members=None, op=dist.ReduceOp.SUM):
        """all-reduce `make(rank)` a.rounds times on `group`; the result must equal expect(members)"""
        if members is not None and rank not in members:
            return
        say("all-reduce %s" % name)
        us = []
        for _ in range(a.rounds):
            t = make(rank).to(coll_dev)
            if coll_dev.type == "cuda":
                torch.cuda.synchronize()
            t0 = time.time()
            dist.all_reduce(t, op=op, group=group)
            if coll_dev.type == "cuda":
                torch.cuda.synchronize()
            us.append((time.time() - t0) * 1e6)
            got = t.cpu()
            want = expect(members if members is not None else list(range(world)))
            if not torch.equal(got, want):
                bad = int((got != want).sum())
                raise RuntimeError("%s: %d of %d elements differ after the all-reduce (first: got %r, want %r)"
                                   % (name, bad, got.numel(), got.flatten()[0].item(), want.flatten()[0].item()))
        us = np.asarray(us[1:] if len(us) > 1 else us)
        report["messages"][name] = {"bytes": int(make(0).numel() * make(0).element_size()), "calls": int(a.rounds),
                                    "us_min": round(float(us.min()), 1), "us_median": round(float(np.median(us)), 1),
                                    "us_max": round(float(us.max()), 1)}

    try:
        K, S, B = 20, 4, 7
        n_stats = K * (1 + S + S * S) + 5
        n_rows = K * (3 * B + 2 + S + S * S) + 1
        ar = lambda m: torch.arange(m, dtype=torch.float64)
        # every rank contributes (rank + 1) * [0, 1, 2, ...]: the sum is known in closed form, element by element
        timed_allreduce("E-step statistics (%d float64)" % n_stats, lambda r: (r + 1) * ar(n_stats),
                        lambda mem: sum(q + 1 for q in mem) * ar(n_stats))
        timed_allreduce("M-step rows (%d float64)" % n_rows, lambda r: (r + 1) * ar(n_rows),
                        lambda mem: sum(q + 1 for q in mem) * ar(n_rows))
        # the label gather: each byte is written by exactly one rank (position mod world), the others add zeros
        nb = 1 << 20
        pos = torch.arange(nb) % max(world, 1)
        timed_allreduce("label gather (%d bytes)" % nb, lambda r: ((pos == r).to(torch.uint8) * torch.tensor(r + 1, dtype=torch.uint8)),
                        lambda mem: (pos + 1).to(torch.uint8))
        timed_allreduce("max-reduce (bench.py's elapsed time)", lambda r: torch.tensor([float(r)], dtype=torch.float64),
                        lambda mem: torch.tensor([float(max(mem))], dtype=torch.float64), op=dist.ReduceOp.MAX)
        say("barrier")
        dist.barrier()

        # ---- the tile sub-groups of each workload at this world size ------------------------------------------------
        from phylo_hmrf_amd import tiles, workloads
        for wl in [w for w in a.workloads.split(",") if w]:
            say("plan %s at %d ranks" % (wl, world))
            blocks_def, S_w, K_w, nn, desc = workloads.workload(wl)
            units = tiles.plan(list(blocks_def), world, a.split_above)
            owner = tiles.assign(units, world)
            groups = []
            for bi, (H, W, diag) in enumerate(blocks_def):
                mine = [(i, u) for i, u in enumerate(units) if u["block"] == bi]
                if len(mine) <= 1:
                    continue
                owners = sorted(set(int(owner[i]) for i, _ in mine))
                say("new_group for block %d of %s: ranks %s (%d tiles)" % (bi, wl, owners, len(mine)))
                grp = dist.new_group(ranks=owners) if len(owners) > 1 else None       # collective: EVERY rank, block order
                # a lockstep round's payload: per tile 128 counters + (unary, pair) energy + viol + the two boundary rows
                per_tile = 128 + 3 + 2 * ((W + 7) // 8)
                nel = per_tile * len(mine)
                groups.append((bi, owners, len(mine), nel))
                if grp is not None:
                    ai = lambda m: torch.arange(m, dtype=torch.int64)
                    timed_allreduce("%s block %d tile round (%d int64, ranks %s)" % (wl, bi, nel, owners),
                                    lambda r, nel=nel: (r + 1) * ai(nel),
                                    lambda mem, nel=nel: sum(q + 1 for q in mem) * ai(nel), group=grp, members=owners)
            report["tile_groups"][wl] = {"units": len(units), "split_blocks": len(groups),
                                         "groups": [{"block": g[0], "ranks": g[1], "tiles": g[2], "int64_per_round": g[3]} for g in groups],
                                         "units_per_rank": [int(sum(1 for o in owner if o == r)) for r in range(world)]}
        say("final barrier")
        dist.barrier()
    except Exception as err:           # name the step; the launcher shows the traceback of the first rank that dies
        sys.stderr.write("[selftest rank %d] FAILED in step: %s\n  %s: %s\n" % (rank, step[0], type(err).__name__, err))
        sys.stderr.flush()
        raise
    ok = torch.tensor([1.0], dtype=torch.float64, device=coll_dev)
    dist.all_reduce(ok)
    report["ranks_that_finished"] = int(ok.item())
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(report), flush=True)
    return 0 if report["ranks_that_finished"] == world else 1


if __name__ == "__main__":
    sys.exit(main())
