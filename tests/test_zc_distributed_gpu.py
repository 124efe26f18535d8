"""The exchange code on the RCCL ("nccl") backend with device tensors.  One GPU is all a test
box has, so the world has a single rank: this checks that RCCL initialises in this image and that
every collective the multi-GPU path issues (barrier, all_reduce, all_gather_into_tensor on device
buffers, from a side stream too) runs; the multi-rank logic is covered over gloo on the CPU."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def world_of_one():
    """One RCCL process group for the module (a second initialisation in one process is not needed
    by anything and is not what a rank does)."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dev
    dist.destroy_process_group()


def test_exchange_on_rccl_world_of_one(world_of_one):
    pkg = importlib.import_module("2d_object_detection_amd")
    d, syn, util = pkg.distributed, pkg.synthetic, pkg.util
    dev = world_of_one
    dist.barrier()
    t = torch.tensor([3.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 3.5
    rows = torch.arange(80, dtype=torch.float32, device=dev).view(10, 8)
    got, counts = d.all_gather_rows(rows, torch.tensor([7], dtype=torch.int32, device=dev))
    assert counts == [7] and torch.equal(got, rows[:7])
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        got2, c2 = d.all_gather_rows(rows, 4)
    torch.cuda.synchronize()
    assert c2 == [4] and torch.equal(got2, rows[:4])
    pg = d.PaddedRowGather(6, 8, dev)
    pg.rows[:5].copy_(rows[:5])
    pg.gather(torch.tensor([5], dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    got3, c3 = pg.unpack()
    assert c3 == [5] and torch.equal(got3, rows[:5])
    # the sharded engine end to end: world of one == the plain engine
    heads = [h.to(dev) for h in syn.make_heads(4, 416, 5, seed=3)]
    post = d.ShardedYoloV3Post(4, 416, 5, 0.5, 0.4, device=dev)
    want = util.YoloV3Post(4, 416, 5, 0.5, 0.4, device=dev).detect(heads)
    assert np.array_equal(post.detect(heads).cpu().numpy(), want.cpu().numpy())


def test_set_gather_behind_the_multi_engine_on_two_streams(world_of_one):
    """The schedule bench.py times at N > 1: two engines of three sets each write straight into
    the payload of their PaddedSetGather, every launch runs on its engine's stream with ONE
    all-gather enqueued behind it on that stream, nothing waits on the host until the end; the
    last launch is partly filled.  After every pair of launches the buffers are unpacked and
    compared with the plain engine on the same heads."""
    pkg = importlib.import_module("2d_object_detection_amd")
    d, syn, util = pkg.distributed, pkg.synthetic, pkg.util
    dev = world_of_one
    S, F, B, cap = 3, 2, 2, 2048
    sets = [[h.to(dev) for h in syn.make_heads(B, 416, 5, seed=40 + k)] for k in range(8)]
    single = util.YoloV3Post(B, 416, 5, 0.5, 0.4, device=dev)
    want = [single.detect(heads).cpu().numpy().copy() for heads in sets]
    assert 0 < max(len(w) for w in want) <= cap
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
    gathers = [d.PaddedSetGather(S, cap, 8, dev) for _ in range(F)]
    engines = [util.YoloV3PostMulti(S, B, 416, 5, 0.5, 0.4, device=dev, out=g.rows) for g in gathers]
    torch.cuda.synchronize()
    plan = [(0, 3), (3, 3), (6, 2), (1, 3)]        # (first set, sets in the launch); launch j -> engine j % F
    for j0 in range(0, len(plan), F):
        for j in range(j0, min(j0 + F, len(plan))):
            first, n = plan[j]
            k = j % F
            _, count = engines[k].run([sets[(first + t) % 8] for t in range(n)], streams[k])
            with torch.cuda.stream(streams[k]):
                gathers[k].gather(count, n)
        torch.cuda.synchronize()
        for j in range(j0, min(j0 + F, len(plan))):
            first, n = plan[j]
            rows, counts = gathers[j % F].unpack()
            assert len(rows) == 1 and len(counts) == 1            # one rank
            assert counts[0] == [len(want[(first + t) % 8]) for t in range(n)] + [0] * (S - n)
            for t in range(n):
                assert np.array_equal(rows[0][t].cpu().numpy(), want[(first + t) % 8]), (j, t)
