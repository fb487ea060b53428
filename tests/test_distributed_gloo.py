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


# ---- BASELINE config 4: ego (rank 0) vs round-robin partners (rank 1 + k), the routing protocol with stand-in agents ----------
class _StubPolicy:
    device = "cpu"


class _StubEgo:
    """what RoundRobinEgoRank needs from a VecOnPolicyAgent"""

    def __init__(self, E):
        self.model = type("M", (), dict(policy=_StubPolicy()))()
        self.E, self.t, self.rewards, self.learned = E, 0, [], 0

    def bind_stream(self):
        pass

    def get_action(self, obs):
        self.t += 1
        return ((th.arange(self.E) + self.t) % 3).to(th.int32)

    def update(self, reward, done):
        self.rewards.append(reward.clone())

    def learn_from_buffer(self):
        self.learned += 1


class _StubPartner:
    """what RoundRobinPartnerRank needs from a RaggedVecOnPolicyAgent: logs every callback"""

    def __init__(self, E, k):
        self.model = type("M", (), dict(policy=_StubPolicy()))()
        self.E, self.k, self.acts, self.upds = E, k, [], []

    def full(self):
        return False

    def get_action(self, obs, mask):
        self.acts.append((obs.clone(), mask.clone()))
        return th.full((self.E, 1), self.k, dtype=th.int32)

    def update(self, reward, done, mask):
        self.upds.append((reward.clone(), done.clone(), mask.clone()))


def _env_step_cpu(joint, partnerid, base, done, reward_out, alt_out, next_block, K, bonus):
    """torch statement of ph_roundrobin_env_step (CPU protocol test only)"""
    e = th.arange(joint.shape[1])
    a_alt = joint[1 + partnerid.long(), e]
    reward_out.copy_(base + bonus * (joint[0] == a_alt).float())
    alt_out.copy_(a_alt)
    partnerid.copy_(th.where(done != 0, (partnerid + 1) % K, partnerid))
    next_block[:, 0], next_block[:, 1], next_block[:, 2] = partnerid.float(), reward_out, done


def _rr_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank))
    from pantheonrl_amd import dist as pdist, roundrobin as rr
    assert pdist.init_from_env("gloo")
    K, E, T, D = world - 1, 6, 5, 4
    g = th.Generator().manual_seed(0)
    obs_alt = th.randn((T, E, D), generator=g)
    data = type("D", (), dict(T=T, E=E, obs=th.randn((T, E, D), generator=g), rewards=th.randn((T, E), generator=g),
                              dones=(th.rand((T, E), generator=g) < 0.4).float()))()
    if rank == 0:
        side = rr.RoundRobinEgoRank(_StubEgo(E), data, obs_alt, K, bonus=0.5, env_step=_env_step_cpu)
        side.run_iteration()
        # the partner of every environment follows (id + 1) % K at that environment's own dones, starting from 1 % K
        pid, want = th.full((E,), 1 % K, dtype=th.int32), []
        for t in range(T):
            want.append(pid.clone())
            pid = th.where(data.dones[t] != 0, (pid + 1) % K, pid)
        assert th.equal(side.partner_trace, th.stack(want))
        # partner k always plays k here, the ego (e + t) % 3: the shared reward is base + bonus * [equal]
        for t in range(T):
            ego_act = (th.arange(E) + t + 1) % 3
            assert th.allclose(side.ego.rewards[t], data.rewards[t] + 0.5 * (ego_act == want[t]).float())
        assert side.ego.learned == 1
        out.put((0, side.partner_trace.tolist()))
    else:
        k = rank - 1
        agent = _StubPartner(E, k)
        side = rr.RoundRobinPartnerRank(agent, k, K, E, D, T)
        side.run_iteration()
        assert len(agent.acts) == T and len(agent.upds) == T - 1
        for t, (o, m) in enumerate(agent.acts):
            assert th.equal(o, obs_alt[t])                     # the routed partner-seat observations
        out.put((rank, [m.tolist() for _, m in agent.acts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_round_robin_routing_protocol_gloo(world):
    """ego rank + (world - 1) partner ranks over gloo on the CPU with stand-in agents: the routing block, the per-environment
    round-robin partner ids and the action hand-back (the -m gpu test replays the real agents' buffers through MultiAgentEnv)"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rr_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(out.get(timeout=5) for _ in range(world))
    trace = np.asarray(got[0])                                  # (T, E) partner id per step and environment
    for k in range(world - 1):                                  # partner k recorded exactly where the trace says k
        assert np.array_equal(np.asarray(got[1 + k]), (trace == k).astype(np.uint8))
