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


def ranks_sharing_device(device, group=None) -> int:
    """how many ranks of the group run on this rank's GPU (1 on a real multi-GPU node; > 1 when ranks time-slice one device, as
    the one-GPU test boxes do) -- the persistent rollout's residency bound and the in-kernel wait bounds need it.  A collective:
    every rank of the group calls it."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 1
    world = dist.get_world_size(group)
    try:                                      # local work only: the collective below is reached by every rank
        props = th.cuda.get_device_properties(device)
        me = f"{os.uname().nodename}:{getattr(props, 'uuid', th.device(device).index)}"
    except Exception:  # noqa: BLE001
        me = None                             # "unknown": counted as sharing with everybody, on every rank alike
    ids = [None] * world
    dist.all_gather_object(ids, me, group=group)
    if any(v is None for v in ids):
        return world                          # the same (pessimistic) answer everywhere
    return max(1, sum(1 for v in ids if v == me))


_generation = {"p2p": 0, "rccl": 0}   # per-process rendezvous counters: every attach uses fresh store keys (ranks attach in
                                        # lockstep, so the n-th attach of every rank meets under the same key)


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
        self.p2p = None             # _native.PhP2P once the peer-to-peer route is attached (attach_p2p)
        self._p2p_slots = None      # [parity] -> (n_seats, n_envs) int32 tensor view of this rank's receive area
        self._p2p_keep = []
        self.attach_generation = 0  # advanced by every attach_p2p call (all ranks call it together): keys cached verdicts

    # -- peer-to-peer route ---------------------------------------------------------------------------------------------
    def attach_p2p(self, ctx, epoch_word: th.Tensor, n_steps: int, timeout_s: Optional[float] = None) -> bool:
        """Map every rank's fine-grained receive area through HIP IPC (handles travel through the process group's store)
        and build the ph_p2p descriptor.  `epoch_word` is the device word the engine advances once per iteration
        (ph_rng_epoch_advance); stamps are epoch * n_steps + t + 1.  Returns False if anything fails.
        `timeout_s` bounds ONE in-kernel wait for a peer's word (default PH_P2P_TIMEOUT_S or 10 s: a bound for lost peers, far
        above any healthy hand-off).  The bound is per device SHARE: when k ranks run on this rank's GPU (`ranks_on_device`; 1 on a
        real node, k on the one-GPU test boxes) a waiting kernel holds the device while the peer it waits for may not be scheduled,
        and the peer can be k - 1 scheduler turns away, so the bound is k times the base."""
        import ctypes as C
        self.attach_generation += 1
        if timeout_s is None:
            timeout_s = float(os.environ.get("PH_P2P_TIMEOUT_S", "10")) * max(int(getattr(self, "ranks_on_device", 1)), 1)

        from . import _native as nat
        if not self.local.is_cuda or self.world > nat.PH_MAX_RANKS:
            return False
        try:
            count = self.local.numel()
            slot = self.world * count * 4
            # stamp-in-band words of the fused route: one slot per step of an iteration.  A rank only waits for its partner's
            # rank, so it may run up to world-1 steps ahead of another one; a slot must not be reused inside an iteration
            ll_slot = self.world * count * 8
            # (the persistent exchange rollout alternates between two halves of 2 T slots, ph_selfplay_rollout_persistent)
            ll_slots = max(2 * int(n_steps), 4)
            ll_base = 2 * slot + self.world * 8 + 64
            nbytes = ll_base + ll_slots * ll_slot
            base, handle = C.c_void_p(), (C.c_ubyte * 64)()
            nat.check(ctx.lib.ph_p2p_alloc(ctx.handle, nbytes, C.byref(base), handle))
            bases = [None] * self.world
            bases[self.rank] = base.value
            if self.world > 1:
                store = dist.distributed_c10d._get_default_store()
                _generation["p2p"] += 1          # a key is written once: a peer can never pick up a previous attach's handle
                gen = _generation["p2p"]
                store.set(f"pantheonrl_amd/p2p/{gen}/{self.rank}", bytes(handle))
                for p in range(self.world):
                    if p == self.rank:
                        continue
                    peer = (C.c_ubyte * 64).from_buffer_copy(store.get(f"pantheonrl_amd/p2p/{gen}/{p}"))
                    mapped = C.c_void_p()
                    nat.check(ctx.lib.ph_p2p_open(ctx.handle, peer, C.byref(mapped)))
                    bases[p] = mapped.value
            x = nat.PhP2P()
            x.world, x.rank, x.count, x.T = self.world, self.rank, count, int(n_steps)
            for p in range(self.world):
                x.joint[0][p], x.joint[1][p] = bases[p], bases[p] + slot
                x.flags[p] = bases[p] + 2 * slot
                x.ll[p] = bases[p] + ll_base
            x.ll_slots = ll_slots
            x.epoch = epoch_word.data_ptr()
            x.error = base.value + 2 * slot + self.world * 8
            x.timeout_cycles = int(timeout_s * 1e8)
            own = base.value

            class _View:   # zero-copy torch view of a raw device range
                def __init__(self, ptr, shape):
                    self.__cuda_array_interface__ = {"shape": shape, "typestr": "<i4", "data": (ptr, False), "version": 2}
            self._p2p_slots = [th.as_tensor(_View(own + par * slot, (self.n_seats, self.n_envs)), device=self.local.device)
                               for par in (0, 1)]
            self._p2p_error = th.as_tensor(_View(x.error, (8,)), device=self.local.device)   # four u64 as eight i32
            self._p2p_keep = [epoch_word, bases]
            self.p2p, self.native_ctx = x, ctx
            return True
        except Exception as exc:  # noqa: BLE001
            import sys
            print(f"[pantheonrl_amd.dist] peer-to-peer exchange unavailable ({exc})", file=sys.stderr)
            self.p2p = None
            return False

    # -- route selection -----------------------------------------------------------------------------------------------------
    def setup(self, ctx, epoch_word: th.Tensor, n_steps: int, route: Optional[str] = None) -> str:
        """Pick how the per-step all-gather is carried: "torch" (torch.distributed call per step), "rccl" (engine-side
        ncclAllGather), "p2p" (direct stores into IPC-mapped peer memory) or "auto": with more than one rank, time the
        engine-side RCCL route against the peer-to-peer route on this very node (after checking that the peer-to-peer
        route delivers the right bytes) and keep the faster one; every rank takes the same decision.  Idempotent."""
        import time
        if getattr(self, "route", None) in ("torch", "rccl", "p2p"):
            return self.route
        route = route or getattr(self, "requested_route", None) or ("p2p" if getattr(self, "want_p2p", False) else "torch")
        log = {}
        if self.local.is_cuda:
            ctx.set_stream(th.cuda.current_stream(self.local.device).cuda_stream)
        self.ranks_on_device = self._ranks_on_device()
        if route == "auto":
            # one rank: the peer-to-peer words mapped onto this rank itself -- what the one-launch exchange rollout runs on
            route = "p2p" if (self.world == 1 and self.local.is_cuda) else ("rccl" if self.world == 1 else "measure")
        if route in ("rccl", "measure") and self.native_ctx is None:
            attached = self._everyone(self.attach_native(ctx)) if self.world > 1 else self.attach_native(ctx)
            verified = False
            if attached:
                # the engine-side all-gather must reproduce torch.distributed's on a known pattern before it is trusted
                # Only the native call sits inside the try: a rank whose native gather throws must still take part in the
                # torch.distributed gather and the verdict all-reduce below, or the other ranks block in mismatched collectives.
                pattern = th.arange(self.local.numel(), dtype=th.int32, device=self.local.device).view_as(self.local)
                self.local.copy_(pattern * 7 + 1000003 * (self.rank + 1))
                got = None
                try:
                    got = self.gather_inplace().clone()
                    if self.local.is_cuda:
                        th.cuda.synchronize(self.local.device)
                except Exception:  # noqa: BLE001
                    got = None
                want = self._torch_gather()           # every rank, unconditionally
                verified = self._everyone(got is not None and bool(th.equal(got, want)))
            self._rccl_verified = verified
            log["rccl_verified"] = verified
            if not verified:
                self.native_ctx = None
                if route == "rccl":
                    route = "torch"
        if route == "measure":
            def timed(step_fn, k=32):
                th.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                for t in range(k):
                    step_fn(t)
                th.cuda.synchronize()
                return (time.perf_counter() - t0) / k
            vdev = self.local.device if dist.get_backend(self.group) == "nccl" else "cpu"
            baseline = "rccl" if self.native_ctx is not None else "torch"

            everyone = self._everyone
            log["rccl_us"] = 1e6 * timed(lambda t: self.gather_inplace())
            T_test = max(int(n_steps), 32)
            ok = everyone(self.attach_p2p(ctx, epoch_word, T_test))
            if ok:
                pattern = th.arange(self.local.numel(), dtype=th.int32, device=self.local.device).view_as(self.local)
                want = th.cat([pattern + 100003 * (r + 1) for r in range(self.world)])
                self.local.copy_(pattern + 100003 * (self.rank + 1))
                good = True
                for step, in_band in ((0, False), (1, True)):   # stamp flags, then the fused launch's stamp-in-band words
                    try:
                        got = self.p2p_step(step, in_band=in_band).clone()
                        th.cuda.synchronize()
                        good = good and bool(th.equal(got, want)) and self.p2p_timeouts() == 0
                    except Exception:  # noqa: BLE001
                        good = False
                    dist.barrier()
                ok = everyone(good)
            if ok:
                log["p2p_us"] = 1e6 * timed(lambda t: self.p2p_step(2 + t, in_band=True), k=30)
                ok = everyone(self.p2p_timeouts() == 0)
            times = th.tensor([log.get("p2p_us", 1e9), log["rccl_us"]], device=vdev)
            dist.all_reduce(times, op=dist.ReduceOp.MAX)          # the slowest rank's view
            # (the timed form is two extra launches per step; in the rollout the exchange is folded into the step launch)
            use_p2p = ok and times[0].item() < 2.0 * times[1].item()
            log.update(p2p_ok=ok, chosen="p2p" if use_p2p else baseline)
            epoch_word += 1          # stamps of the measurement must not satisfy the first real iteration's waits
            th.cuda.synchronize()
            dist.barrier()
            if use_p2p:
                route = "p2p"        # (the descriptor keeps T_test >= n_steps: stamps stay unique and monotonic)
            else:
                self.p2p, self._p2p_slots, route = None, None, baseline
                if baseline == "torch":
                    self.native_ctx = None
        elif route == "p2p":
            if not self._everyone(self.attach_p2p(ctx, epoch_word, n_steps)):
                self.p2p, self._p2p_slots = None, None
                route = "rccl" if (self.native_ctx is not None or self._everyone(self.attach_native(ctx))) else "torch"
        self.route, self.route_log = route, log
        return route

    def _ranks_on_device(self) -> int:
        return ranks_sharing_device(self.local.device, self.group) if self.local.is_cuda else 1

    def _torch_gather(self) -> th.Tensor:
        """all-gather of `self.local` through torch.distributed alone (the route of last resort and the yardstick the native
        routes are verified against); returns a fresh (n_seats, n_envs) tensor"""
        out = th.empty_like(self.joint)
        if self.world == 1:
            out.copy_(self.local)
        elif self.local.is_cuda and dist.get_backend(self.group) == "gloo":
            host = th.empty(self.joint.shape, dtype=self.joint.dtype)
            dist.all_gather_into_tensor(host, self.local.cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, self.local, group=self.group)
        return out

    def _everyone(self, flag: bool) -> bool:
        """the same verdict on every rank, whatever happened locally (a rank that failed must take the others with it)"""
        if self.world == 1:
            return bool(flag)
        vdev = self.local.device if dist.get_backend(self.group) == "nccl" else "cpu"
        v = th.tensor([1.0 if flag else 0.0], device=vdev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN, group=self.group)
        return bool(v.item() > 0.5)

    def verify_route(self, last_slot: th.Tensor) -> bool:
        """After a real iteration: does the joint action this rank consumed for the last step equal what torch.distributed
        gathers from the same local actions?  (`self.local` still holds the last step's actions.)  All ranks get the same
        answer; a peer-to-peer timeout also counts as a failure."""
        ok = True
        try:                                      # local work only: nothing in here is a collective
            if self.local.is_cuda:
                th.cuda.synchronize(self.local.device)
            ok = self.p2p_timeouts() == 0
        except Exception:  # noqa: BLE001
            ok = False
        want = self._torch_gather()               # every rank, unconditionally: the collective sequence is identical on all ranks
        try:
            ok = ok and bool(th.equal(last_slot, want))
        except Exception:  # noqa: BLE001
            ok = False
        return self._everyone(ok)

    def demote(self) -> str:
        """give up the native route in use: peer-to-peer -> engine-side RCCL if that one verified, else torch.distributed"""
        was = getattr(self, "route", "torch")
        self.p2p = None
        self._p2p_slots = None
        if was == "p2p" and self.native_ctx is not None and getattr(self, "_rccl_verified", False):
            self.route = "rccl"
        else:
            self.native_ctx, self.route = None, "torch"
        self.route_log = dict(getattr(self, "route_log", {}), demoted_from=was, chosen=self.route)
        return self.route

    def joint_slot(self, parity: int) -> th.Tensor:
        """the (n_seats, n_envs) joint-action buffer holding steps of this parity (one buffer unless peer-to-peer)"""
        return self._p2p_slots[parity & 1] if self.p2p is not None else self.joint

    def p2p_timeouts(self) -> int:
        return int(self._p2p_error[0].item()) if self.p2p is not None else 0

    def p2p_timeout_record(self) -> Optional[dict]:
        """what the FIRST timed-out wait of this rank was waiting for (csrc/ph_launch.h: p2p_note_timeout), or None"""
        if self.p2p is None or self.p2p_timeouts() == 0:
            return None
        w = [int(v) & 0xFFFFFFFF for v in self._p2p_error.cpu().tolist()]
        kinds = {1: "stamp-in-band word polled by a value tail", 2: "stamp-in-band word polled by the unpack kernel",
                 3: "stamp flag of the push / wait pair", 4: "round-robin stamp"}
        index = w[3]
        rec = {"rank": self.rank, "kind": kinds.get(w[2] & 0xFF, w[2] & 0xFF), "step": w[2] >> 8,
               "want_stamp": w[4], "seen_stamp": w[7], "seen_value": w[6], "timeouts": self.p2p_timeouts()}
        if (w[2] & 0xFF) in (1, 2):
            rec["seat"], rec["row"] = index // max(self.n_envs, 1), index % max(self.n_envs, 1)
            T = int(self.p2p.T)
            rec["seen_is"] = ("nothing yet" if w[7] == 0 else
                              f"iteration {(w[7] - 1) // T} step {(w[7] - 1) % T}") + f"; wanted iteration {(w[4] - 1) // T} step {(w[4] - 1) % T}"
        else:                                   # 64-bit stamps: the low words
            rec["source_rank"], rec["seen_stamp"], rec["seen_value"] = index, w[6], None
        return rec

    def p2p_step(self, t: int, in_band: bool = False) -> th.Tensor:
        """one exchange of step t outside the fused rollout loop (tests, self-test): push + wait with stamp flags, or -- the
        protocol of the fused step launch -- stamp-in-band words + unpack"""
        import ctypes as C

        from . import _native as nat
        ctx = self.native_ctx
        if in_band:
            nat.check(ctx.lib.ph_p2p_ll_push(ctx.handle, C.byref(self.p2p), self.local.data_ptr(), int(t)))
            nat.check(ctx.lib.ph_p2p_ll_unpack(ctx.handle, C.byref(self.p2p), int(t)))
        else:
            nat.check(ctx.lib.ph_p2p_push(ctx.handle, C.byref(self.p2p), self.local.data_ptr(), int(t)))
            nat.check(ctx.lib.ph_p2p_wait(ctx.handle, C.byref(self.p2p), int(t)))
        return self.joint_slot(t)

    def attach_native(self, ctx) -> bool:
        """Create the engine-side RCCL communicator on `ctx` (a _native.Context): rank 0 draws the unique id, the store of
        the default process group carries it.  Returns False (and keeps the torch.distributed route) when that is not
        possible: CPU tensors, a gloo group sharing one GPU, or librccl missing."""
        import ctypes as C

        from . import _native as nat
        if not self.local.is_cuda:
            return False
        # (PANTHEON_RCCL_WITH_GLOO=1: attempt the communicator although the rendezvous group is gloo -- the unique id travels through
        # the store either way.  tests/scripts/rccl_two_rank.py uses it to make first contact with ncclCommInitRank at nranks = 2 on
        # whatever the box has: two GPUs -> a real two-rank all-gather; one GPU -> RCCL's duplicate-device refusal and the demotion)
        if self.world > 1 and (self.group is not None or
                               (dist.get_backend() != "nccl" and os.environ.get("PANTHEON_RCCL_WITH_GLOO") != "1")):
            return False
        try:
            ident = (C.c_ubyte * 128)()
            if self.world > 1:
                store = dist.distributed_c10d._get_default_store()
                _generation["rccl"] += 1
                key = f"pantheonrl_amd/rccl_id/{_generation['rccl']}"
                if self.rank == 0:
                    nat.check(ctx.lib.ph_comm_unique_id(ident))
                    store.set(key, bytes(ident))
                else:
                    ident = (C.c_ubyte * 128).from_buffer_copy(store.get(key))
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
