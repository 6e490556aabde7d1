"""world_size-2 tests of the N>1 path: static image split -> independent per-rank processing -> gather of the result
records.  On the CPU (gloo) the per-image worker is the oracle (tests may use it); the `-m gpu` variant runs the PRODUCT --
HipContext through the C ABI -- in both ranks (they share the one GPU of the test box; on a node every rank has its own,
exactly as bench.py launches them with backend "nccl" = RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pngloss_amd as P
from pngloss_amd import shard as S
from tests import util as U

SIZES = [(40, 12), (64, 48), (17, 5), (33, 9), (70, 20), (8, 8), (50, 31)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, gpu=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shards = S.lpt_partition([w * h for (w, h) in SIZES], world)
        local = []
        if gpu:
            # the product path: one device-resident batch per rank through the C ABI
            imgs = [P.synth_rgba(SIZES[i][0], SIZES[i][1], i % 6, i) for i in shards[rank]]
            dev = [torch.from_numpy(a.copy()).cuda() for a in imgs]
            filt = [torch.zeros(a.shape[0], dtype=torch.uint8, device="cuda") for a in imgs]
            hctx = P.HipContext(0)
            res = hctx.run([(d.data_ptr(), f.data_ptr(), a.shape[1], a.shape[0]) for d, f, a in zip(dev, filt, imgs)], 19, 2)
            torch.cuda.synchronize()
            hctx.close()
            assert all(r["status"] == 0 for r in res)
            for i, d, f in zip(shards[rank], dev, filt):
                local.append(dict(index=i, rank=rank, out=P.fnv1a64(d.cpu().numpy()), filters=P.fnv1a64(f.cpu().numpy()),
                                  pixels=SIZES[i][0] * SIZES[i][1]))
        for i in ([] if gpu else shards[rank]):
            w, h = SIZES[i]
            out, f = U.run_port(P.synth_rgba(w, h, i % 6, i), 19, 2)
            local.append(dict(index=i, rank=rank, out=P.fnv1a64(out), filters=P.fnv1a64(f), pixels=w * h))
        dist.barrier()
        allrec = S.gather_records(local)
        t = torch.tensor([float(sum(r["pixels"] for r in local))])
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if rank == 0:
            q.put((allrec, float(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_shard_and_gather_through_the_hip_path():
    _two_ranks(gpu=True)


def test_two_rank_gloo_shard_and_gather():
    _two_ranks(gpu=False)


def _two_ranks(gpu):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, gpu)) for r in range(world)]
    for p in procs:
        p.start()
    allrec, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r["index"] for r in allrec] == list(range(len(SIZES)))
    assert {r["rank"] for r in allrec} == {0, 1}
    assert total == sum(w * h for (w, h) in SIZES)
    # every record equals what a single process computes: sharding does not change results
    for r in allrec:
        w, h = SIZES[r["index"]]
        out, f = U.run_port(P.synth_rgba(w, h, r["index"] % 6, r["index"]), 19, 2)
        assert r["out"] == P.fnv1a64(out) and r["filters"] == P.fnv1a64(f)


def test_gather_records_single_process_passthrough():
    recs = [dict(index=2), dict(index=0), dict(index=1)]
    assert [r["index"] for r in S.gather_records(recs)] == [0, 1, 2]
