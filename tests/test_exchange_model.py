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
