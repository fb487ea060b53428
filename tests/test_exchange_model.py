"""A host-side model of the stamp-in-band peer-to-peer exchange (include/pantheon_hip.h: ph_p2p, `ll` area).

Every rank runs T steps per iteration.  At step t it (a) consumes its partner's word of step t-1 -- it polls the slot
(t-1) mod S of its OWN receive area until the stamp matches -- and (b) stores its own word of step t, stamped, into slot
t mod S of EVERY rank's area.  A rank waits for its partner's rank only, so along a chain of partners a rank can run ahead
of another one by up to world-1 steps.  The model explores the worst-case schedule and shows why the engine insists on
`ll_slots >= T` (ph_abi.hip: check_p2p): with fewer slots than the reachable skew + 2 a producer overwrites a word that a
slow consumer has not read yet, and that consumer then polls for a stamp that never comes."""
import itertools

import pytest


def run(world: int, T: int, slots: int, order):
    """-> None if every rank finishes, else (rank, step) of a consumer stuck on an overwritten word.  `order` yields rank ids:
    the scheduler (a rank that cannot make progress when scheduled just yields)."""
    area = [[[None] * world for _ in range(slots)] for _ in range(world)]   # area[dst][slot][src] = stamp
    step = [0] * world                                                        # next step of every rank
    partner = [(r + 1) % world for r in range(world)]                         # round-robin pairing, one seat per rank
    idle = 0
    for r in order:
        t = step[r]
        if t == T:
            idle += 1
        elif t >= 1 and area[r][(t - 1) % slots][partner[r]] != t - 1:
            have = area[r][(t - 1) % slots][partner[r]]
            if have is not None and have > t - 1:
                return r, t                      # the word was overwritten by a later step: this poll can only time out
            idle += 1
        else:
            for dst in range(world):
                area[dst][t % slots][r] = t
            step[r] = t + 1
            idle = 0
        if all(s == T for s in step):
            return None
        if idle > 4 * world:
            raise AssertionError("model deadlocked without an overwrite")
    raise AssertionError("schedule exhausted")


def eager_first(world):
    """a schedule that lets low ranks run as far ahead as their partners allow before higher ranks move at all"""
    while True:
        for r in reversed(range(world)):          # the chain r -> r+1: rank world-1 first gives rank 0 the longest lead
            for _ in range(world):
                yield r


@pytest.mark.parametrize("world", [2, 4, 8])
def test_one_slot_per_step_never_overwrites_an_unread_word(world):
    T = 16
    assert run(world, T, slots=T, order=itertools.islice(eager_first(world), 100000)) is None


def test_three_slots_are_enough_for_two_ranks_only():
    T = 16
    assert run(2, T, slots=3, order=itertools.islice(eager_first(2), 100000)) is None
    stuck = run(4, T, slots=3, order=itertools.islice(eager_first(4), 100000))
    assert stuck is not None, "four ranks skew by three steps along the partner chain: slot t mod 3 is overwritten"


# ---- several iterations: the pairing changes per iteration, the word slots are reused across iterations ---------------------
# What the rollouts add on top of the per-step hand-off (pantheonrl_amd/vec.py: FusedSelfPlayRollout.run_iteration):
#   * the pairing of iteration k is partner_of(seat, k) = (seat + 1 + k mod (world - 1)) mod world  (dist.ActionExchange);
#   * at the END of an iteration every rank unpacks the words of step T - 1 of ALL ranks (ph_selfplay_rollout_p2p /
#     ph_selfplay_rollout_persistent) -- the one point per iteration where a rank waits for everybody;
#   * two slot disciplines: the launch-per-step form uses slot t mod S in every iteration, the one-launch form alternates between
#     two halves, slot (k & 1) * T + t (csrc/ph_launch.h: p2p_persistent_slot).
# A poll is "lost" when the slot holds a LATER stamp than the one waited for (overwritten before it was read), and the model
# reports a standstill when no rank can move (on the device: every poll runs into its timeout).
def run_iterations(world, T, slots, iters, forms, schedule):
    def slot(form, k, t):
        return t % slots if form == "step" else (k & 1) * T + t

    def stamp(k, t):
        return k * T + t + 1

    area = [[[0] * world for _ in range(max(slots, 2 * T))] for _ in range(world)]   # area[dst][slot][src] = stamp
    pos = [[0, 0] for _ in range(world)]                                             # (iteration, step); step == T: the unpack

    def state(r):
        """'done', 'ready', 'waiting', or ('overwritten', src)"""
        k, t = pos[r]
        if k == iters:
            return "done"
        partner = (r + 1 + k % (world - 1)) % world if world > 1 else r
        need = [(partner, t - 1)] if 1 <= t < T else ([(p, T - 1) for p in range(world)] if t == T else [])
        verdict = "ready"
        for src, ts in need:
            have = area[r][slot(forms[r], k, ts)][src]
            if have > stamp(k, ts):
                return ("overwritten", src)
            if have != stamp(k, ts):
                verdict = "waiting"
        return verdict

    for r in schedule:
        st = state(r)
        if isinstance(st, tuple):
            return ("overwritten", r) + tuple(pos[r]) + (st[1],)
        if st == "ready":
            k, t = pos[r]
            if t < T:
                for dst in range(world):
                    area[dst][slot(forms[r], k, t)][r] = stamp(k, t)
                pos[r] = [k, t + 1]
            else:
                pos[r] = [k + 1, 0]
        elif all(state(q) in ("done", "waiting") for q in range(world)):
            if all(p[0] == iters for p in pos):
                return None
            return ("standstill",) + tuple(tuple(p) for p in pos)      # nobody can move: on the device every poll times out
        if all(p[0] == iters for p in pos):
            return None
    raise AssertionError("schedule exhausted")


def random_schedule(world, seed, burst=7):
    import random
    rng = random.Random(seed)
    while True:
        r = rng.randrange(world)
        for _ in range(rng.randint(1, burst * world)):     # a rank keeps the device for a while (process time slices)
            yield r


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("form", ["step", "pers"])
def test_slot_reuse_across_iterations_is_safe_when_every_rank_uses_the_same_form(world, form):
    """config 5's test sizes (T = 16, 2 T slots, three iterations with three different pairings -- rings of 8, 2 x 4 and 8 ranks)
    under the eager schedule and 40 random time-slice schedules: no word is overwritten before it is read, nobody stands still"""
    T, iters = 16, 4
    scheds = [eager_first(world)] + [random_schedule(world, s) for s in range(40)]
    for sch in scheds:
        assert run_iterations(world, T, 2 * T, iters, [form] * world, itertools.islice(sch, 2000000)) is None


def test_ranks_that_disagree_on_the_rollout_form_stand_still_on_the_first_odd_iteration():
    """Why the verdict "one launch or one launch per step" must be the same on every rank (vec.FusedSelfPlayRollout.persistent_ok
    all-reduces it since round 4; it used to be taken from rank-local inputs): the two forms put iteration 1's words into
    different slots, so a rank in the other form polls a slot its partner never writes -- on the device every such poll runs
    into its timeout and the run is declared invalid, which is the one failure mode of the 8-rank test this model can produce."""
    world, T = 8, 16
    forms = ["pers"] * world
    forms[3] = "step"
    out = run_iterations(world, T, 2 * T, 3, forms, itertools.islice(random_schedule(world, 1), 2000000))
    assert out is not None and out[0] == "standstill"
    assert all(k == 1 for k, _ in out[1:]), out      # iteration 0 is indistinguishable (both forms use slots 0 .. T-1 there)
