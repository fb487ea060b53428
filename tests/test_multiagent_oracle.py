"""not-gpu: oracle/multiagent_oracle.py (an independent restatement of the reference's MultiAgentEnv step / reset control flow for a
2-player simultaneous game with round-robin partners) and the product's pantheonrl_amd.common.SimultaneousEnv must drive a random
scripted game identically: same observations / rewards / dones / `_partnerid` to the ego, same get_action / update call sequence
to every partner.  Neither is derived from the other; tests/scripts/roundrobin_ranks.py replays device runs through both."""
import numpy as np

from oracle.multiagent_oracle import RoundRobinSimultaneousOracle
from pantheonrl_amd.common import Agent, SimultaneousEnv


def _drive(env, T):
    out, ob = [], env.reset()
    for _ in range(T):
        ob2, r, d, info = env.step(0)
        out.append((float(ob2), float(r), bool(d), list(info["_partnerid"])))
        ob = env.reset() if d else ob2
        if d:
            out.append(("reset", float(ob)))
    return out


def test_oracle_and_product_wrappers_agree_call_for_call():
    rng = np.random.default_rng(0)
    T, K = 400, 3
    dones = rng.random(T) < 0.2
    rew = rng.standard_normal((T, 2)).astype(np.float32)

    class Rec(Agent):
        def __init__(self):
            self.log = []

        def get_action(self, obs, record=True):
            self.log.append(("act", float(np.asarray(obs.obs))))
            return 1

        def update(self, reward, done):
            self.log.append(("upd", float(reward), bool(done)))

    class Game:
        def __init__(self):
            self.g = 0

        def multi_reset(self):
            return float(self.g), float(-self.g)

        def multi_step(self, a0, a1):
            t = self.g
            self.g += 1
            return (float(self.g), float(-self.g)), (float(rew[t, 0]), float(rew[t, 1])), bool(dones[t]), {}

    class Product(Game, SimultaneousEnv):
        def __init__(self):
            SimultaneousEnv.__init__(self)
            Game.__init__(self)

    pa, pb = [Rec() for _ in range(K)], [Rec() for _ in range(K)]
    product = Product()
    for a in pa:
        product.add_partner_agent(a)
    oracle = RoundRobinSimultaneousOracle(Game(), pb)
    assert _drive(product, T) == _drive(oracle, T)
    assert [a.log for a in pa] == [b.log for b in pb]
    assert all(len(a.log) > 0 for a in pa)            # every partner was played against
    # the first episode is played with partner 1, not 0: the id advances at every reset, the first included (multiagentenv.py:224)
    assert pb[1].log[0][0] == "act" and pb[1].log[1] == ("upd", 0.0, False)
