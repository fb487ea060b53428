"""not-gpu: the agent-per-GPU exchange layer with world_size 2 over gloo (the N>1 path of bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank))
    from pantheonrl_amd import dist as pdist
    assert pdist.init_from_env("gloo")
    ex = pdist.ActionExchange(agents_local=2, n_envs=16, device="cpu")
    assert (ex.world, ex.rank, ex.n_seats, ex.bytes_per_step) == (world, rank, 4, 4 * 16 * 4)
    results = []
    for step in range(3):
        local = [th.full((16,), 100 * rank + 10 * i + step, dtype=th.int32) for i in range(2)]
        joint = ex.gather(local)
        results.append(joint.clone())
    # every seat's actions arrive at every rank, rank-major seat order
    for step, joint in enumerate(results):
        expect = th.tensor([[100 * r + 10 * i + step] * 16 for r in range(world) for i in range(2)], dtype=th.int32)
        assert th.equal(joint, expect), (rank, step)
    # round-robin pairing (multiagentenv.py:118-125 generalised): never self, cycles through every other seat
    for seat in range(ex.n_seats):
        partners = [ex.partner_of(seat, r) for r in range(ex.n_seats - 1)]
        assert seat not in partners and sorted(partners) == [s for s in range(ex.n_seats) if s != seat]
    assert ex.seat(1) == rank * 2 + 1
    out.put((rank, [int(j.sum()) for j in results]))
    dist.barrier()
    dist.destroy_process_group()


def test_action_exchange_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(out.get(timeout=5) for _ in range(2))
    assert got[0] == got[1]   # both ranks hold the same joint action


def test_single_process_exchange_is_identity():
    from pantheonrl_amd import dist as pdist
    assert pdist.init_from_env() is False
    ex = pdist.ActionExchange(agents_local=2, n_envs=4, device="cpu")
    joint = ex.gather([th.arange(4, dtype=th.int32), th.arange(4, 8, dtype=th.int32)])
    assert joint.tolist() == [[0, 1, 2, 3], [4, 5, 6, 7]] and ex.partner_of(0) == 1 and ex.partner_of(1) == 0
