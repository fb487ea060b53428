"""Agent-per-GPU exchange layer.

In the reference every agent<->env hand-off is an in-process call inside MultiAgentEnv._get_actions /
_update_players (pantheonrl/common/multiagentenv.py:149-170).  With one learner per GPU the same hand-off becomes:
every rank contributes the actions of its local agents for the current SimultaneousEnv step, every rank receives the
joint action.  Learners are independent (reference README.md:6): there is NO gradient or parameter exchange, hence no
all-reduce anywhere -- only this KB-sized, latency-bound all-gather per environment step.  On GPUs it is issued by the
engine itself (`ph_all_gather_i32`: ncclAllGather of RCCL on the engine's stream, communicator bootstrapped through
torch.distributed's store) so that a step costs two native enqueues and no Python collective call; torch.distributed
(backend "nccl" == RCCL; "gloo" on CPU for the world_size-2 tests) is the fallback and the rendezvous.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch as th
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> bool:
    """initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); returns False for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return False
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if th.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    return True


class ActionExchange:
    """all-gather of per-step actions: local (A_local, E) int32 -> joint (world * A_local, E) int32.

    Seats are numbered rank-major: global seat g = rank * A_local + local index.  `partner_of(g)` implements the
    round-robin pairing the reference applies per episode (multiagentenv.py:118-125): at pairing round r, seat g plays
    with seat (g + 1 + r mod (n_seats-1)) mod n_seats -- kept in Python, like the reference's partner selection.
    """

    def __init__(self, agents_local: int, n_envs: int, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.agents_local, self.n_envs = agents_local, n_envs
        self.n_seats = self.world * agents_local
        self.local = th.zeros((agents_local, n_envs), dtype=th.int32, device=device)
        self.joint = th.zeros((self.n_seats, n_envs), dtype=th.int32, device=device)
        self.bytes_per_step = self.joint.numel() * 4
        self.native_ctx = None      # engine context holding the RCCL communicator (attach_native)

    def attach_native(self, ctx) -> bool:
        """Create the engine-side RCCL communicator on `ctx` (a _native.Context): rank 0 draws the unique id, the store of
        the default process group carries it.  Returns False (and keeps the torch.distributed route) when that is not
        possible: CPU tensors, a gloo group sharing one GPU, or librccl missing."""
        import ctypes as C

        from . import _native as nat
        if not self.local.is_cuda:
            return False
        if self.world > 1 and (self.group is not None or dist.get_backend() != "nccl"):
            return False
        try:
            ident = (C.c_ubyte * 128)()
            if self.world > 1:
                store = dist.distributed_c10d._get_default_store()
                if self.rank == 0:
                    nat.check(ctx.lib.ph_comm_unique_id(ident))
                    store.set("pantheonrl_amd/rccl_id", bytes(ident))
                else:
                    ident = (C.c_ubyte * 128).from_buffer_copy(store.get("pantheonrl_amd/rccl_id"))
            elif os.environ.get("PANTHEON_FORCE_RCCL", "0") == "1":   # one-rank communicator: exercises the RCCL plumbing
                nat.check(ctx.lib.ph_comm_unique_id(ident))
            else:
                ident = None       # single process: the engine's all-gather degenerates to a device copy
            if ident is not None:
                nat.check(ctx.lib.ph_comm_init(ctx.handle, ident, self.world, self.rank))
        except Exception as exc:  # noqa: BLE001 -- any failure here just means "use the torch.distributed route"
            import sys
            print(f"[pantheonrl_amd.dist] native RCCL exchange unavailable ({exc}); using torch.distributed",
                  file=sys.stderr)
            return False
        self.native_ctx = ctx
        return True

    def seat(self, local_index: int) -> int:
        return self.rank * self.agents_local + local_index

    def partner_of(self, seat: int, pairing_round: int = 0) -> int:
        if self.n_seats < 2:
            return seat
        return (seat + 1 + pairing_round % (self.n_seats - 1)) % self.n_seats

    def gather(self, local_actions: List[th.Tensor]) -> th.Tensor:
        """local_actions[i]: (E,) or (E,1) int32 of local agent i -> joint (n_seats, E).  Enqueued on the current
        stream; no host synchronisation."""
        for i, a in enumerate(local_actions):
            self.local[i].copy_(a.reshape(-1))
        return self.gather_inplace()

    def gather_inplace(self) -> th.Tensor:
        """all-gather of `self.local` (already filled in place, e.g. by the policy-forward kernels) -> `self.joint`."""
        if self.native_ctx is not None:
            from . import _native as nat
            nat.check(self.native_ctx.lib.ph_all_gather_i32(self.native_ctx.handle, self.local.data_ptr(),
                                                            self.joint.data_ptr(), self.local.numel()))
        elif self.world == 1:
            self.joint.copy_(self.local)
        elif self.local.is_cuda and dist.get_backend(self.group) == "gloo":
            # test-only route (two ranks sharing one GPU cannot form an RCCL communicator): stage through the host
            host = th.empty(self.joint.shape, dtype=self.joint.dtype)
            dist.all_gather_into_tensor(host, self.local.cpu(), group=self.group)
            self.joint.copy_(host)
        else:
            dist.all_gather_into_tensor(self.joint, self.local, group=self.group)
        return self.joint
