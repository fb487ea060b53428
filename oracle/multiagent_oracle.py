"""TEST INFRASTRUCTURE (oracle): a CPU restatement of the control flow PantheonRL wraps around a 2-player simultaneous game --
`MultiAgentEnv.reset / step / _get_actions / _update_players` with round-robin partner resampling
(/root/reference/pantheonrl/common/multiagentenv.py:113-125,149-243) and `SimultaneousEnv.n_step / n_reset` (:395-409) --
written for ONE purpose: replaying what the device did in BASELINE config 4 (ego vs K round-robin partners) through the
reference's order of operations WITHOUT going through the product's own `pantheonrl_amd.common.SimultaneousEnv`.

Only tests/ may import this module.  It is plain Python on scalars; nothing here is shipped or measured.

What is restated (reference line in brackets):
  * reset(): the partner id advances FIRST -- `(id + 1) % K`, also at the very first reset [:118-125, :224]; then the game is
    reset, `should_update = False`, `total_rews = [0, 0]`, `ego_moved = False` [:225-228].  In a simultaneous game the ego is
    among the players of every step, so the "play until the ego moves" loop [:230-238] never runs.
  * step(a): the partner of this episode is asked for its action on the partner-seat observation [:152-157]; the first time it
    acts in an episode it is first handed the reward accrued before it moved, `update(total_rews[1], False)` [:158-160]; the
    game transition [:193]; `info['_partnerid']` = the id list [:194]; the partner receives `update(rews[1], done)` [:163-167];
    `total_rews += rews` [:169-170]; the ego's reward is `rews[0]` once it has moved, `total_rews[0]` on its first move of
    the episode [:198-199]; on `done` the PREVIOUS ego observation is returned [:203-205], otherwise the new one [:210-212].
"""
from __future__ import annotations

from typing import Any, List, Tuple


class ObservationOracle:
    """`Observation(obs)` of the reference (observation.py:7-25) as far as this path uses it: the raw observation in `.obs`"""

    def __init__(self, obs):
        self.obs = obs
        self.state = None
        self.action_mask = None


class RoundRobinSimultaneousOracle:
    """game: an object with `multi_reset() -> (obs0, obs1)` and `multi_step(a0, a1) -> ((obs0, obs1), (r0, r1), done, info)`;
    partners: the K agents of seat 1 (`get_action(Observation) -> action`, `update(reward, done)`)."""

    def __init__(self, game, partners: List[Any]):
        self.game, self.partners = game, list(partners)
        self.partnerid = 0                      # MultiAgentEnv.__init__: partnerids = [0]
        self.should_update = False
        self.total_rews = [0.0, 0.0]
        self.ego_moved = False
        self._obs: Tuple[Any, Any] = (None, None)
        self._old_ego_obs = None

    def reset(self):
        self.partnerid = (self.partnerid + 1) % len(self.partners)          # resample_round_robin
        obs0, obs1 = self.game.multi_reset()
        self._obs = (ObservationOracle(obs0), ObservationOracle(obs1))
        self.should_update = False
        self.total_rews = [0.0, 0.0]
        self.ego_moved = False
        self._old_ego_obs = self._obs[0]
        return self._obs[0].obs                                             # ego_extractor default: lambda obs: obs.obs

    def step(self, ego_action):
        agent = self.partners[self.partnerid]
        alt_action = agent.get_action(self._obs[1])
        if not self.should_update:
            agent.update(self.total_rews[1], False)
        self.should_update = True
        (obs0, obs1), rews, done, info = self.game.multi_step(ego_action, alt_action)
        self._obs = (ObservationOracle(obs0), ObservationOracle(obs1))
        info = dict(info)
        info["_partnerid"] = [self.partnerid]
        if self.should_update:
            agent.update(rews[1], done)
        self.total_rews[0] += rews[0]
        self.total_rews[1] += rews[1]
        ego_rew = rews[0] if self.ego_moved else self.total_rews[0]
        self.ego_moved = True
        if done:
            return self._old_ego_obs.obs, ego_rew, done, info
        self._old_ego_obs = self._obs[0]
        return self._obs[0].obs, ego_rew, done, info
