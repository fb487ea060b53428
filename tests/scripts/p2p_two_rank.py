"""Two (or more) ranks exchanging per-step actions through the peer-to-peer route (HIP IPC-mapped fine-grained receive areas,
system-scope stamps).  Launched by tests/test_gpu_parity.py with torchrun; ranks may share one GPU (gloo rendezvous)."""
import os
import sys

import numpy as np
import torch as th
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pantheonrl_amd import _native as nat  # noqa: E402
from pantheonrl_amd import dist as pdist  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev_index = int(os.environ.get("LOCAL_RANK", "0")) % th.cuda.device_count()
th.cuda.set_device(dev_index)
dist.init_process_group("gloo")
device = th.device("cuda", dev_index)
E, A_LOCAL, T = 256, 2, 8
ctx = nat.Context(dev_index)
stream = th.cuda.Stream()
epoch = th.zeros(1, dtype=th.int64, device=device)
ex = pdist.ActionExchange(A_LOCAL, E, device)
assert ex.attach_p2p(ctx, epoch, T, timeout_s=5.0), "attach_p2p failed"
dist.barrier()
with th.cuda.stream(stream):
    ctx.set_stream(stream.cuda_stream)
    for it in range(3):
        for t in range(T):
            ex.local.copy_(th.arange(A_LOCAL * E, dtype=th.int32, device=device).view(A_LOCAL, E) + 1000 * rank + 7 * t + it)
            joint = ex.p2p_step(t, in_band=bool(it % 2)).clone()
            stream.synchronize()
            want = np.concatenate([np.arange(A_LOCAL * E, dtype=np.int32).reshape(A_LOCAL, E) + 1000 * r + 7 * t + it
                                   for r in range(world)])
            assert np.array_equal(joint.cpu().numpy(), want), (rank, it, t)
        epoch += 1          # what ph_rng_epoch_advance does once per iteration
        stream.synchronize()
        dist.barrier()      # nobody starts the next iteration's pushes before everybody has checked this one
assert ex.p2p_timeouts() == 0, ex.p2p_timeouts()
dist.barrier()

# ---- phase 2: the fused route (push / stamp / wait inside the step launch) under the real rollout driver ----------------
from pantheonrl_amd import PPO, spaces as sp  # noqa: E402
from pantheonrl_amd.vec import FusedSelfPlayRollout, SyntheticRollouts, VecOnPolicyAgent  # noqa: E402

E2, T2, BONUS = 128, 6, 0.25
obs_space, act_space = sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6)
env = type("S", (), dict(observation_space=obs_space, action_space=act_space, _is_dummy_space_env=True))()
agents, datas = [], []
for i in range(A_LOCAL):
    seed = 10 * rank + i
    m = PPO("MlpPolicy", env, n_steps=T2, n_envs=E2, batch_size=E2 * T2 // 2, n_epochs=1, seed=seed)
    agents.append(VecOnPolicyAgent(m))
    datas.append(SyntheticRollouts(obs_space, E2, T2, 400, seed, m.device))
def rewards_as_the_joint_actions_imply(roll_, ex_, it):
    """every rank's actions of this iteration, via the rendezvous group, to rebuild the expected rewards on the host"""
    mine = th.stack([a.model.rollout_buffer.actions[..., 0].cpu() for a in agents])          # (A_LOCAL, T, E)
    everyone = [th.zeros_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    seats = th.cat(everyone).numpy()                                                          # (n_seats, T, E)
    for i, (a, d) in enumerate(zip(agents, datas)):
        seat = ex_.seat(i)
        partner = ex_.partner_of(seat, it)
        expect = d.rewards.cpu().numpy() + BONUS * (seats[seat] == seats[partner])
        got = a.model.rollout_buffer.rewards.cpu().numpy()
        bad = got != expect.astype(np.float32)
        assert not bad.any(), (rank, it, i, roll_.last_rollout_mode, "mismatching rows per step", bad.sum(1).tolist(),
                               "timeouts", ex_.p2p_timeouts())


# launch per step (push / stamp / wait inside every step launch), then the whole rollout as ONE launch per rank with the
# hand-off in-kernel (ph_selfplay_rollout_persistent): the ranks' persistent launches run side by side on the shared GPU and
# wait for each other's words step by step; four iterations so that both halves of the word slots are reused
for persistent, n_it in ((False, 2), (None, 4)):
    ex2 = pdist.ActionExchange(A_LOCAL, E2, device)
    ex2.requested_route = "p2p"
    with th.cuda.stream(stream):
        roll = FusedSelfPlayRollout(agents, datas, ex2, stream, bonus=BONUS, update_graphs=False, persistent=persistent)
        assert ex2.route == "p2p", ex2.route
        assert ex2.ranks_on_device == (world if th.cuda.device_count() == 1 else ex2.ranks_on_device)
        for it in range(n_it):
            dist.barrier()
            roll.run_iteration(it)
            stream.synchronize()
            assert roll.last_rollout_mode == ("p2p" if persistent is False else "persistent"), roll.last_rollout_mode
            rewards_as_the_joint_actions_imply(roll, ex2, it)
    assert ex2.p2p_timeouts() == 0, ex2.p2p_timeouts()
    assert roll.route_checked and ex2.route == "p2p"       # the post-iteration verification ran and kept the route
    dist.barrier()

# ---- phase 3: a route that fails its verification after the first real iteration is dropped on every rank, and the run goes on
ex3 = pdist.ActionExchange(A_LOCAL, E2, device)
ex3.requested_route = "p2p"
with th.cuda.stream(stream):
    roll3 = FusedSelfPlayRollout(agents, datas, ex3, stream, bonus=BONUS, update_graphs=False)
    assert ex3.route == "p2p"
    ex3.verify_route = lambda slot: False                  # as if a peer's words had not arrived
    for it in range(2):
        dist.barrier()
        roll3.run_iteration(it)
        stream.synchronize()
        if it == 0:
            assert ex3.route in ("rccl", "torch") and ex3.p2p is None and ex3.route_log.get("demoted_from") == "p2p", ex3.route_log
            continue                                       # (the iteration that exposed the failure is not checked)
        mine = th.stack([a.model.rollout_buffer.actions[..., 0].cpu() for a in agents])
        everyone = [th.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        seats = th.cat(everyone).numpy()
        for i, (a, d) in enumerate(zip(agents, datas)):
            seat = ex3.seat(i)
            expect = d.rewards.cpu().numpy() + BONUS * (seats[seat] == seats[ex3.partner_of(seat, it)])
            assert np.array_equal(a.model.rollout_buffer.rewards.cpu().numpy(), expect.astype(np.float32)), (rank, i)
dist.barrier()
print(f"P2P_OK rank {rank}/{world}", flush=True)
dist.destroy_process_group()
