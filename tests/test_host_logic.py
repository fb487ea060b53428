"""not-gpu: the host-side mirror of pantheonrl.common -- Agent / OnPolicyAgent callbacks, MultiAgentEnv driving,
partner selection, the two in-tree games -- exercised with scripted agents and an oracle-backed model."""
import os

import numpy as np
import pytest
import torch as th

from pantheonrl_amd.common import (Agent, MultiAgentEnv, Observation, OnPolicyAgent, PlayerException,
                                   SimultaneousEnv, StaticPolicyAgent, TurnBasedEnv)
from pantheonrl_amd.common import util
from pantheonrl_amd.envs import make
from pantheonrl_amd.envs.liar import CALL, LiarDefaultAgent, LiarEnv
from pantheonrl_amd.envs.rps import RPSEnv, RPSWeightedAgent, rps_payoff
from pantheonrl_amd.spaces import Box, Discrete, MultiBinary, MultiDiscrete, SpaceException
from tests.stub_model import OraclePPO


class Scripted(Agent):
    """records every callback; plays a fixed action"""

    def __init__(self, action=0):
        self.action, self.log = action, []

    def get_action(self, obs, record=True):
        self.log.append(("act", obs.obs.tolist() if hasattr(obs.obs, "tolist") else obs.obs))
        return self.action

    def update(self, reward, done):
        self.log.append(("upd", reward, done))


# ---- MultiAgentEnv --------------------------------------------------------------------------------------------------
def test_simultaneous_env_drives_partner_callbacks_in_reference_order():
    env = RPSEnv()
    partner = Scripted(action=1)  # paper
    env.add_partner_agent(partner)
    obs = env.reset()
    assert obs.tolist() == [0] and partner.log == []
    obs, rew, done, info = env.step(2)  # scissors beats paper
    assert (rew, done, info["_partnerid"]) == (1.0, True, [0])
    # first move of the episode: get_action, then update(total_rews so far = 0, False), then update(reward, done)
    assert partner.log == [("act", [0]), ("upd", 0, False), ("upd", -1, True)]
    env.reset()
    _, rew, _, _ = env.step(0)          # rock loses to paper
    assert rew == -1.0
    assert np.array_equal(rps_payoff(np.array([0, 1, 2, 0]), np.array([0, 0, 0, 2])), [0, 1, -1, 1])


def test_round_robin_advances_on_every_reset_including_the_first():
    env = RPSEnv()
    partners = [Scripted(a) for a in (0, 1, 2)]
    for p in partners:
        env.add_partner_agent(p)
    seen = []
    for _ in range(7):
        env.reset()
        _, _, _, info = env.step(0)
        seen.append(info["_partnerid"][0])
    assert seen == [1, 2, 0, 1, 2, 0, 1]        # the first episode already uses partner 1
    env.set_resample_policy("random")
    np.random.seed(0)
    ids = set()
    for _ in range(30):
        env.reset()
        ids.add(env.partnerids[0])
    assert ids == {0, 1, 2}
    env.set_partnerid(2)
    assert env.partnerids == [2]
    with pytest.raises(AssertionError):
        env.set_partnerid(3)


def test_player_exceptions():
    with pytest.raises(PlayerException):
        RPSEnv().set_resample_policy("nope")

    class Three(MultiAgentEnv):
        def n_step(self, a):
            return (0,), (Observation(np.zeros(1)),), (0, 0, 0), True, {}

        def n_reset(self):
            return (1,), (Observation(np.zeros(1)),)

    with pytest.raises(PlayerException):
        Three(n_players=3, resample_policy="robin")
    with pytest.raises(PlayerException):
        Three(n_players=3, partners=[[Scripted()]])            # wrong number of seats
    with pytest.raises(PlayerException):
        Three(n_players=3, partners=[[Scripted()], []])        # empty seat list
    env = Three(n_players=3)
    with pytest.raises(PlayerException):
        env.add_partner_agent(Scripted(), player_num=0)        # the ego seat is not a partner seat
    env.add_partner_agent(Scripted(), player_num=2)
    assert len(env.partners[0]) == 0 and len(env.partners[1]) == 1   # D-7: seats do not alias
    env.add_partner_agent(Scripted(), player_num=1)
    with pytest.raises(PlayerException):                       # game ends before the ego ever moves
        env.reset()


def test_turn_based_env_and_first_move_reward_handover():
    class Count(TurnBasedEnv):
        """ego and partner alternate; every move pays (1, 10); game ends after 4 moves"""

        def __init__(self):
            super().__init__(probegostart=1.0)
            self.observation_space, self.action_space = Discrete(8), Discrete(2)
            self.moves = 0

        def _move(self):
            self.moves += 1
            return np.array([self.moves]), (1, 10), self.moves >= 4, {}

        def ego_step(self, action):
            return self._move()

        def alt_step(self, action):
            return self._move()

        def multi_reset(self, egofirst):
            self.moves = 0
            return np.array([0])

    env = Count()
    p = Scripted()
    env.add_partner_agent(p)
    assert env.reset().tolist() == [0]
    obs, rew, done, _ = env.step(0)           # ego move (1) then partner move (2)
    assert (obs.tolist(), rew, done) == ([2], 2.0, False)
    # partner's first action is credited the 10 it accrued during the ego's move (multiagentenv.py:158-159)
    assert p.log == [("act", [1]), ("upd", 10, False), ("upd", 10, False)]
    obs, rew, done, _ = env.step(0)           # ego move (3), partner move (4) ends the game
    assert (rew, done) == (2.0, True) and obs.tolist() == [2]   # D-8: previous ego obs on done
    assert p.log[-3:] == [("upd", 10, False), ("act", [3]), ("upd", 10, True)]

    env2 = Count()
    env2.probegostart = 0.0                   # partner moves first inside reset()
    p2 = Scripted()
    env2.add_partner_agent(p2)
    assert env2.reset().tolist() == [1]
    assert p2.log == [("act", [0]), ("upd", 0, False), ("upd", 10, False)]
    _, rew, _, _ = env2.step(0)
    assert rew == 1 + 1 + 1                   # ego_moved False: total_rews so far (1 from reset) + this round's 2


# ---- OnPolicyAgent ---------------------------------------------------------------------------------------------------
def test_onpolicy_agent_buffer_contents_and_train_before_act():
    env = RPSEnv()
    model = OraclePPO(env, n_steps=4, batch_size=4, n_epochs=1)
    agent = OnPolicyAgent(model)
    assert agent._last_episode_starts == [True] and list(model.ep_info_buffer) == [{"r": 0, "l": 0}]
    obs = Observation(np.array([0]))
    script = [(1.0, True), (0.5, False), (-1.0, True), (2.0, True)]
    for rew, done in script:
        a = agent.get_action(obs)
        assert a in (0, 1, 2)
        agent.update(0, False)      # multiple updates per action add up, last done wins (agents.py:44-47)
        agent.update(rew, done)
    buf = model.rollout_buffer
    assert buf.full and model.train_calls == 0                      # D-3: nothing trained yet
    assert buf.rewards.ravel().tolist() == [1.0, 0.5, -1.0, 2.0]
    assert buf.episode_starts.ravel().tolist() == [1, 1, 0, 1]     # [True] initially, then the previous done
    last_value = float(agent.values.reshape(-1)[0])
    agent.get_action(obs)                                            # 5th action: GAE + train + reset first
    assert model.train_calls == 1 and agent.iteration == 1 and agent.n_steps == 1 and buf.pos == 1
    t = model.trained_on[0]
    # D-1: the bootstrap is V(o_{T-1}) with dones = last done (True) -> A_{T-1} = r - V
    assert abs(t["advantages"].ravel()[-1] - (2.0 - t["values"].ravel()[-1])) < 1e-6
    assert abs(t["values"].ravel()[-1] - last_value) < 1e-7
    # episode bookkeeping: three finished episodes + the one in progress
    eps = list(model.ep_info_buffer)
    assert [e["r"] for e in eps[:3]] == [1.0, -0.5, 2.0] and [e["l"] for e in eps[:3]] == [1, 2, 1]
    # record=False still advances n_steps but writes no row (D-4)
    agent.get_action(obs, record=False)
    assert agent.n_steps == 2 and buf.pos == 1


def test_rps_ppo_vs_ppo_preset_object_graph():
    """trainer.py preset-1 graph (env, altenv=getDummyEnv(1), partner OnPolicyAgent(PPO(altenv))) on the oracle model:
    partner trains at the NEXT get_action after its buffer fills."""
    env = make("RPS-v0")
    altenv = env.getDummyEnv(1)
    assert altenv is env
    partner = OnPolicyAgent(OraclePPO(altenv, n_steps=64, batch_size=64, n_epochs=1), tb_log_name="alt")
    env.add_partner_agent(partner)
    env.reset()
    for _ in range(200):
        _, _, done, _ = env.step(np.random.randint(3))
        assert done
        env.reset()
    assert partner.num_timesteps == 200 and partner.model.train_calls == 3   # after steps 64, 128, 192
    assert len(partner.model.ep_info_buffer) == 100


def test_static_policy_agent_and_util_helpers():
    env = make("LiarsDice-v0")
    model = OraclePPO(env, n_steps=8)
    agent = StaticPolicyAgent(model.policy)
    act = agent.get_action(Observation(np.zeros(30)))
    assert act.shape == (2,) and 0 <= act[0] < 7 and 0 <= act[1] < 12
    agent.update(1.0, True)
    assert util.get_space_size(Box(-1, 1, (5,))) == 5 and util.get_space_size(Discrete(4)) == 1
    assert util.get_space_size(MultiBinary(3)) == 3 and util.get_space_size(MultiDiscrete([2, 3])) == 2
    assert util.calculate_space(Discrete(4), 3).nvec.tolist() == [4, 4, 4]
    assert util.calculate_space(MultiDiscrete([2, 3]), 2).nvec.tolist() == [2, 3, 2, 3]
    assert util.calculate_space(Box(-1, 1, (2,)), 2).shape == (4,) and util.calculate_space(MultiBinary(2), 3).n == 6
    with pytest.raises(SpaceException):
        util.get_space_size(object())
    assert util.get_default_obs(env) == [0] * 30
    clipped = util.clip_actions(np.array([[5.0]]), type("P", (), {"action_space": Box(-1, 1, (1,))})())
    assert clipped.tolist() == [[1.0]]


# ---- Liar's Dice integer rules ---------------------------------------------------------------------------------------------
def test_liars_dice_rules():
    np.random.seed(0)
    env = LiarEnv(probegostart=1.0)
    env.add_partner_agent(LiarDefaultAgent())
    obs = env.reset()
    assert obs.shape == (30,) and obs[:6].sum() == 6 and obs[6:].tolist() == [6, 0] * 12
    # opening "call" is sanitised to [0, 0]; a non-raising bid is a call
    assert env.sanitize_action(np.array([6, 3])) == [0, 0]
    env.history = [2, 4]
    assert env.sanitize_action(np.array([1, 4])) == CALL and env.sanitize_action(np.array([6, 9])) == CALL
    assert env.sanitize_action(np.array([1, 5])) == [1, 5]
    env.egohand, env.althand = [0, 0, 3, 0, 3, 0], [0, 0, 2, 0, 4, 0]
    env.history = [2, 4]       # "five 3s" (count stored minus one): exactly 5 on the table -> not a bluff
    assert env.eval_bluff() is False
    env.history = [2, 5]
    assert env.eval_bluff() is True
    # ego calls a bluff -> ego wins
    _, rews, done, _ = env.player_step(np.array(CALL), True)
    assert (rews, done) == ((1, -1), True)
    env.history = [2, 4]
    _, rews, done, _ = env.player_step(np.array(CALL), True)
    assert (rews, done) == ((-1, 1), True)
    # a full game through the MultiAgentEnv driver terminates with a +-1 reward
    for _ in range(20):
        env.reset()
        total, done, steps = 0, False, 0
        while not done:
            _, r, done, _ = env.step(env.action_space.sample())
            total += r
            steps += 1
        assert total in (1, -1) and steps <= 13
    w = RPSWeightedAgent(r=1, p=0, s=0)
    assert all(w.get_action(None) == 0 for _ in range(5))


# ---- frame stacking (wrappers.py) ----------------------------------------------------------------------------------------
def test_history_queue_and_frame_stack_wrappers():
    from pantheonrl_amd.common.wrappers import HistoryQueue, SimultaneousFrameStack, TurnBasedFrameStack, frame_wrap
    q = HistoryQueue([0, 0], 3)
    assert q.add([1, 2]).tolist() == [1, 2, 0, 0, 0, 0]
    assert q.add([3, 4]).tolist() == [3, 4, 1, 2, 0, 0]
    assert q.add([5, 6]).tolist() == [5, 6, 3, 4, 1, 2]
    assert q.add([7, 8]).tolist() == [7, 8, 5, 6, 3, 4]          # oldest dropped, most recent first (wrappers.py:61-63)
    q.reset()
    assert q.add([9, 9]).tolist() == [9, 9, 0, 0, 0, 0]
    env = frame_wrap(RPSEnv(), 4)
    assert isinstance(env, SimultaneousFrameStack) and env.observation_space.nvec.tolist() == [1, 1, 1, 1]
    partner = Scripted(1)
    env.add_partner_agent(partner)
    assert env.reset().tolist() == [0, 0, 0, 0]
    env.step(0)
    assert partner.log[0] == ("act", [0, 0, 0, 0])
    liar = frame_wrap(LiarEnv(probegostart=1.0), 2)
    assert isinstance(liar, TurnBasedFrameStack) and liar.observation_space.shape == (60,)
    liar.add_partner_agent(LiarDefaultAgent())
    np.random.seed(1)
    first = liar.reset()
    assert first.shape == (60,) and first[30:].tolist() == [0] * 30 and first[:6].sum() == 6   # default obs = zeros
    obs, _, done, _ = liar.step(np.array([0, 0]))
    if not done:
        assert obs[30:].tolist() == first[:30].tolist()   # the previous ego observation slid to the second slot


# ---- trainer.py object graph: argument surface ---------------------------------------------------------------------------
def test_trainer_cli_surface_and_presets():
    from pantheonrl_amd import trainer
    p = trainer.build_parser()
    args = p.parse_args(["RPS-v0", "PPO", "PPO", "DEFAULT", "--seed", "7", "--preset", "1", "-t", "123",
                         "--alt-config", '{"n_steps": 64}', "--alt-config", '{"r": 2}'])
    assert (args.env, args.ego, args.alt, args.total_timesteps) == ("RPS-v0", "PPO", ["PPO", "DEFAULT"], 123)
    assert args.alt_config == [{"n_steps": 64}, {"r": 2}]
    trainer.input_check(args)
    args = trainer.preset(args, 1)
    assert args.tensorboard_log == "logs" and args.tensorboard_name == "RPS-v0-PPOPPO-7"
    assert args.ego_save == "models/RPS-v0-PPO-ego-7" and args.alt_save == "models/RPS-v0-PPO-alt-7"
    trainer.parse_cli(["RPS-v0", "ADAP_MULT", "ADAP", "--share-latent", "--alt-config", "{}"])   # trainer.py:32-34,67-71
    for bad in (["RPS-v0", "ADAP_MULT", "PPO", "--share-latent"], ["RPS-v0", "PPO", "BC"], ["OvercookedMultiEnv-v0", "PPO", "PPO"],
                ["RPS-v0", "PPO", "PPO", "--share-latent"], ["RPS-v0", "ADAP", "PPO", "--share-latent"],
                ["RPS-v0", "SAC", "PPO"]):
        with pytest.raises(trainer.EnvException):
            trainer.parse_cli(bad)
    with pytest.raises(trainer.EnvException):
        trainer.parse_cli(["RPS-v0", "PPO", "PPO", "--alt-config", "{}", "--alt-config", "{}"])   # two configs for one partner
    # --share-latent: ADAP everywhere, partners inherit the ego's context settings (trainer.py:65-89)
    a = trainer.parse_cli(["RPS-v0", "ADAP", "ADAP", "--share-latent", "--ego-config", '{"context_size": 4}'])
    assert a.alt_config == [{"context_size": 4, "context_sampler": "l2"}] and a.ego_config["context_sampler"] == "l2"
    with pytest.raises(trainer.EnvException):
        trainer.parse_cli(["RPS-v0", "ADAP", "ADAP", "--share-latent", "--alt-config", '{"context_size": 2}'])
    # DEFAULT partners need no GPU: the graph builds up to the ego
    a = p.parse_args(["LiarsDice-v0", "PPO", "DEFAULT", "--framestack", "3"])
    a.alt_config = [{}]
    env, altenv = trainer.generate_env(a)
    assert env.observation_space.shape == (90,) and altenv is env
    agent = trainer.gen_partner("DEFAULT", {}, altenv, None, a, 0)
    assert isinstance(agent, LiarDefaultAgent)
    with pytest.raises(trainer.EnvException):
        trainer.gen_partner("DEFAULT", {"r": 1}, altenv, None, a, 0)
    assert isinstance(trainer.gen_partner("DEFAULT", {"r": 1, "p": 0, "s": 0}, RPSEnv(), None, a, 0), RPSWeightedAgent)


def test_three_player_aec_style_game_like_the_pettingzoo_adapter():
    """n_players = 3, one player acts per n_step (the shape of PettingZooAECWrapper.n_step, pettingzoo.py:54-103, and of
    BASELINE config 5): ego is seat 0, partners in seats 1 and 2 are resampled at random per episode."""
    class Ring(MultiAgentEnv):
        """players act 0,1,2,0,1,2,...; each action pays its mover `action` and everybody else 0; 2 full rounds"""

        def __init__(self):
            super().__init__(ego_ind=0, n_players=3)
            self.observation_space, self.action_space = Discrete(10), Discrete(5)
            self.t = 0

        def n_reset(self):
            self.t = 0
            return (0,), (Observation(np.array([0])),)

        def n_step(self, actions):
            mover = self.t % 3
            rews = [0.0, 0.0, 0.0]
            rews[mover] = float(actions[0])
            self.t += 1
            done = self.t >= 6
            return ((self.t % 3,), (Observation(np.array([self.t])),), tuple(rews), done, {})

    env = Ring()
    assert env.resample_partner == env.resample_random        # "default" policy for > 2 players (multiagentenv.py:134-139)
    a1, a2, b2 = Scripted(1), Scripted(2), Scripted(4)
    env.add_partner_agent(a1, player_num=1)
    env.add_partner_agent(a2, player_num=2)
    env.add_partner_agent(b2, player_num=2)
    np.random.seed(3)
    seen = set()
    for _ in range(12):
        obs = env.reset()
        assert obs.tolist() == [0]
        seat2 = env.partners[1][env.partnerids[1]]
        seen.add(env.partnerids[1])
        a1.log.clear(); seat2.log.clear()
        obs, rew, done, info = env.step(3)          # ego acts (t=0), then seats 1 and 2 (t=1,2)
        assert (obs.tolist(), rew, done) == ([3], 3.0, False) and info["_partnerid"] == env.partnerids
        # seat 1: acts on obs [1]; hand-over update(0, False); its own move's reward; then seat 2's move (reward 0)
        assert a1.log == [("act", [1]), ("upd", 0.0, False), ("upd", 1.0, False), ("upd", 0.0, False)]
        assert seat2.log == [("act", [2]), ("upd", 0.0, False), ("upd", float(seat2.action), False)]
        obs, rew, done, _ = env.step(2)             # second round ends the game at t=6
        assert done and rew == 2.0 and obs.tolist() == [3]      # D-8: previous ego observation on done
        assert a1.log[-1] == ("upd", 0.0, True) and seat2.log[-1] == ("upd", float(seat2.action), True)
    assert seen == {0, 1}
    with pytest.raises(PlayerException):
        env.set_resample_policy("robin")


# ---- trajectory recorders and the .npy wire format (wrappers.py:82-230, trajsaver.py:130-232) ----------------------------
def test_recorders_and_npy_wire_format(tmp_path):
    from pantheonrl_amd.common import (SimultaneousTransitions, TransitionsMinimal, TurnBasedTransitions,
                                       recorder_wrap)
    from pantheonrl_amd.common.wrappers import SimultaneousRecorder, TurnBasedRecorder
    # simultaneous: RPS, 5 one-step episodes
    env = recorder_wrap(RPSEnv())
    assert isinstance(env, SimultaneousRecorder)
    env.add_partner_agent(Scripted(1))
    for a in (0, 1, 2, 0, 1):
        env.reset()
        env.step(a)
    tr = env.get_transitions()
    assert tr.egoacts.tolist() == [0, 1, 2, 0, 1] and tr.altacts.tolist() == [1] * 5 and tr.flags.tolist() == [1] * 5
    assert tr.egoobs.shape == (5, 1) and len(tr.get_alt_transitions()) == 5
    f = tmp_path / "rps.npy"
    tr.write_transition(f)
    table = np.load(f)
    assert table.shape == (5, 5) and table[:, 1].tolist() == [0, 1, 2, 0, 1] and table[:, -1].tolist() == [1] * 5
    back = SimultaneousTransitions.read_transition(f, env.observation_space, env.action_space)
    assert np.array_equal(back.egoacts.ravel(), tr.egoacts) and np.array_equal(back.flags, tr.flags)
    # turn based: Liar's Dice, the scripted partner against random ego moves
    np.random.seed(4)
    lenv = recorder_wrap(LiarEnv())
    assert isinstance(lenv, TurnBasedRecorder)
    lenv.add_partner_agent(LiarDefaultAgent())
    episodes = 0
    while episodes < 6:
        lenv.reset()
        done = False
        while not done:
            _, _, done, _ = lenv.step(lenv.action_space.sample())
        episodes += 1
    lenv.reset()                                  # a dangling observation that nobody acted on: dropped at export
    tb = lenv.get_transitions()
    assert len(tb.obs) == len(tb.acts) == len(tb.flags) and tb.obs.shape[1] == 30 and tb.acts.shape[1] == 2
    assert int(np.sum(tb.flags >= 2)) == 6        # exactly one game-ending move per episode
    assert set(np.unique(tb.flags)) <= {0, 1, 2, 3}
    ego, alt = tb.get_ego_transitions(), tb.get_alt_transitions()
    assert len(ego) + len(alt) == len(tb.flags) and len(ego) > 0 and len(alt) > 0
    g = tmp_path / "liar.npy"
    tb.write_transition(g)
    assert np.load(g).shape == (len(tb.flags), 33)          # [obs 30 | acts 2 | flag]
    back = TurnBasedTransitions.read_transition(g, lenv.observation_space, lenv.action_space)
    assert np.array_equal(back.obs, tb.obs) and np.array_equal(back.acts, tb.acts) and np.array_equal(back.flags, tb.flags)
    h = tmp_path / "ego.npy"
    ego.write_transition(h)
    again = TransitionsMinimal.read_transition(h, lenv.observation_space, lenv.action_space)
    assert np.array_equal(again.obs, ego.obs) and again[0]["acts"].shape == (2,) and len(again[1:3]) == 2
    with pytest.raises(ValueError):
        TransitionsMinimal(np.zeros((3, 2)), np.zeros((2, 1)))


def test_recording_agent_wrapper_and_offpolicy_refusal(tmp_path):
    """RecordingAgentWrapper (reference agents.py:365-413): delegates both callbacks and keeps every (obs, action) pair;
    the pairs come back as a TransitionsMinimal that survives the .npy round trip.  OffPolicyAgent is outside the
    on-policy path and says so."""
    from pantheonrl_amd.common import OffPolicyAgent, RecordingAgentWrapper, TransitionsMinimal
    env = RPSEnv()
    inner = Scripted(action=2)
    rec = RecordingAgentWrapper(inner)
    env.add_partner_agent(rec)
    for ego_move in (0, 1, 2):
        env.reset()
        env.step(ego_move)
    assert [e[0] for e in inner.log] == ["act", "upd", "upd"] * 3            # both callbacks reach the real agent
    tr = rec.get_transitions()
    assert isinstance(tr, TransitionsMinimal) and len(tr) == 3
    assert tr.obs.tolist() == [[0]] * 3 and tr.acts.tolist() == [2, 2, 2]
    tr.write_transition(str(tmp_path / "rec.npy"))
    back = TransitionsMinimal.read_transition(str(tmp_path / "rec.npy"), env.observation_space, env.action_space)
    assert np.array_equal(back.obs.reshape(-1), tr.obs.reshape(-1)) and np.array_equal(back.acts.reshape(-1), tr.acts)
    with pytest.raises(NotImplementedError, match="on-policy"):
        OffPolicyAgent(object())


def test_games_reproduce_the_hand_worked_traces_and_the_committed_fixture():
    """tests/golden/liar_hand_worked.json: games worked by hand from the reference's liar.py (the independent anchor);
    tests/golden/game_traces.npz: seeded traces written by make_game_traces.py from this restatement (pins drift)."""
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for g in json.load(open(os.path.join(here, "liar_hand_worked.json")))["games"]:
        t = LiarEnv()
        t.history, t.egohand, t.althand = [], list(g["egohand"]), list(g["althand"])
        for s in g["steps"]:
            o, r, d, _ = t.player_step(np.asarray(s["raw"]), s["is_ego"])
            assert np.asarray(o).tolist() == s["obs"] and list(r) == s["rew"] and d == s["done"], (g["note"], s)
    z = np.load(os.path.join(here, "game_traces.npz"))
    assert np.array_equal(rps_payoff(z["rps_ego"], z["rps_alt"]).astype(np.float32), z["rps_ego_reward"])
    E = z["liar_hands"].shape[0]
    turn = z["liar_ego_first"].astype(bool)
    tables = []
    for e in range(E):
        t = LiarEnv()
        t.history, t.egohand, t.althand = [], z["liar_hands"][e, :6].tolist(), z["liar_hands"][e, 6:].tolist()
        tables.append(t)
    for s in range(z["liar_acts"].shape[0]):
        for e in np.nonzero(z["liar_alive"][s])[0]:
            o, r, d, _ = tables[e].player_step(z["liar_acts"][s, e], bool(turn[e]))
            assert np.array_equal(np.asarray(o, np.float32), z["liar_obs"][s, e]) and tuple(r) == tuple(z["liar_rew"][s, e])
            assert bool(d) == bool(z["liar_done"][s, e])
        turn = ~turn


def test_integer_fixtures_are_what_the_reference_code_produces():
    """tests/golden/check_against_reference.py executes the reference's own liar.py / rps.py (from /root/reference, with
    inert stand-ins for the absent `gym` names) on the committed inputs and compares every output: the integer fixtures are
    reference output, not only restatement output.  Build container only -- the reference does not travel to the GPU box."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("check_against_reference", os.path.join(here, "check_against_reference.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    if not os.path.isdir(chk.REFERENCE):
        pytest.skip("the reference tree is not present on this machine")
    liar_ns, rps_ns = chk.load_reference_games()
    assert chk.check_hand_worked(liar_ns) == 24
    z = dict(np.load(os.path.join(here, "game_traces.npz")))
    for k, v in chk.replay_traces(liar_ns, rps_ns, z).items():
        assert np.array_equal(v, z[k]), k
    chk.check_dice(liar_ns)
    # and the product's restatement agrees with the reference on fresh random play (not only on the committed inputs)
    rng = np.random.default_rng(11)
    for _ in range(200):
        hands = rng.integers(0, 7, 12)
        ref = chk._table(liar_ns, hands[:6], hands[6:])
        own = LiarEnv()
        own.history, own.egohand, own.althand = [], hands[:6].tolist(), hands[6:].tolist()
        ego = bool(rng.integers(0, 2))
        for _s in range(14):
            raw = np.asarray([rng.integers(0, 7), rng.integers(0, 12)])
            o1, r1, d1, _ = ref.player_step(raw.copy(), ego)
            o2, r2, d2, _ = own.player_step(raw.copy(), ego)
            assert np.array_equal(np.asarray(o1), np.asarray(o2)) and tuple(r1) == tuple(r2) and bool(d1) == bool(d2)
            if d1:
                break
            ego = not ego


def test_tester_and_bctrainer_cli_surface():
    """tester.py:14-31 / bctrainer.py:26-67: flags and the load-iff-not-DEFAULT rule (no GPU: nothing is constructed)"""
    from pantheonrl_amd import bctrainer, tester, trainer
    a = tester.build_parser().parse_args(["RPS-v0", "PPO", "DEFAULT", "--ego-load", "m/ego", "-t", "7", "--alt-config", '{"r": 2}',
                                          "-f", "3", "--render", "-r", "out.npy"])
    tester.input_check(a)
    assert (a.total_episodes, a.framestack, a.record, a.alt_config, a.ego_config) == (7, 3, "out.npy", {"r": 2}, {"verbose": 1})
    for bad in (["RPS-v0", "PPO", "DEFAULT"], ["RPS-v0", "PPO", "PPO", "--ego-load", "x"],
                ["RPS-v0", "PPO", "DEFAULT", "--ego-load", "x", "--alt-load", "y"], ["RPS-v0", "ModularAlgorithm", "PPO", "--ego-load", "x"]):
        with pytest.raises(trainer.EnvException):
            tester.input_check(tester.build_parser().parse_args(bad))
    b = bctrainer.build_parser().parse_args(["LiarsDice-v0", "demo.npy", "--choose-alt", "-t", "4", "--l2", "0.01", "--save", "c.pt"])
    assert (b.env, b.trajectory, b.choose_alt, b.total_epochs, b.l2, b.save, b.framestack) == ("LiarsDice-v0", "demo.npy", True, 4,
                                                                                             0.01, "c.pt", 1)


def test_lds_swizzles_of_the_split_gradient_kernel_are_conflict_free_under_the_lane_group_model():
    """pl_swz / h2_swz of ph_ppo_split.hip restated as linear maps (scripts/lds_swizzle_search.py) and run through the LDS lane-group
    model that predicted the kernel's measured SQ_LDS_BANK_CONFLICT: operand reads (plain and transposing), the X commit and the
    dZ2 commit are conflict-free, the 8-byte C-layout stores keep their inherent 2-way conflict, H2's head reads are conflict-free."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "lds_swizzle_search.py")
    spec = importlib.util.spec_from_file_location("lds_swizzle_search", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.plane_conflicts(mod.PLANE_SWZ)
    assert res["plain b128 read (x2)"] == 0 and res["transposing b64 read (x16)"] == 0
    assert res["X commit b128 write (x8)"] == 0 and res["dZ2 commit b128 write (x8)"] == 0
    assert res["C-layout b64 write (x16)"] == 64                      # one extra cycle per 16-lane group
    first = mod.plane_conflicts((0b1000, 0b0100, 0b0010))             # the layout the kernel started with
    assert mod.plane_score(first) > 5 * mod.plane_score(res)
    writes, reads = mod.h2_conflicts(mod.H2_SWZ)
    assert reads == 0 and writes == 128
    # the device function, restated: pl_swz(a) = a1 | (a1 ^ a2) << 1 | (a0 ^ a1 ^ a3) << 2 ; h2_swz(r) = (r & 1) | ((r & 2) ? 12 : 0)
    s, f = mod.linear_map(mod.PLANE_SWZ), mod.linear_map(mod.H2_SWZ)
    for a in range(16):
        a0, a1, a2, a3 = a & 1, (a >> 1) & 1, (a >> 2) & 1, (a >> 3) & 1
        assert s(a) == (a1 | ((a1 ^ a2) << 1) | ((a0 ^ a1 ^ a3) << 2))
        assert f(a) == ((a & 1) | (12 if a & 2 else 0))


# ----------------------------------------------------------------------------------------------------------------
# the host step path's Python (agents.py:111-199 as ONE native call per callback), with the native library stubbed
# ----------------------------------------------------------------------------------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stubbed_agent():
    import importlib.util
    spec = importlib.util.spec_from_file_location("host_step_overhead", os.path.join(ROOT, "scripts", "host_step_overhead.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def test_host_step_path_makes_one_native_call_per_callback():
    from pantheonrl_amd.common.observation import Observation
    agent = _stubbed_agent()
    pol, rb = agent.model.policy, agent.model.rollout_buffer
    calls = pol.ctx.lib.calls
    rng = np.random.default_rng(0)
    kept = []
    for t in range(6):
        act = agent.get_action(Observation(rng.standard_normal(62)))        # float64 in: converted while staging
        kept.append(agent.values)
        agent.update(0.25 * t, t == 3)
        assert np.asarray(act).shape == () and rb.pos == t + 1 and agent.n_steps == t + 1
    assert calls == ["ph_policy_act_host", "ph_buffer_add_reward_const"] * 6
    assert agent._last_episode_starts == [False] and len(agent.model.ep_info_buffer) == 2   # one episode ended at t == 3
    assert agent.model.ep_info_buffer[0] == {"r": 0.25 * (0 + 1 + 2 + 3), "l": 4}
    assert agent.model.ep_info_buffer[1] == {"r": 0.25 * (4 + 5), "l": 2}
    # results are copies of the staging arrays, one staging record per observation shape
    assert all(isinstance(v, np.ndarray) and v.shape == (1, 1) for v in kept)
    assert len({v.ctypes.data for v in kept[-2:]}) == 2 and len(pol._host_out) == 1
    hs = pol._host_out[(1, 62)]
    assert hs.rows.dtype == np.float32 and hs.es[0] == 0.0                  # the last step's episode_start was False
    # a numpy scalar reward takes the general entry point (same native call), a per-env array the array form
    agent.update(np.float32(1.0), False)
    rb.n_envs = 1
    assert calls[-1] == "ph_buffer_add_reward_const"


def test_host_step_path_refuses_a_full_buffer_and_restages_on_a_new_shape():
    from pantheonrl_amd import _native as nat
    agent = _stubbed_agent()
    pol, rb = agent.model.policy, agent.model.rollout_buffer
    a, v, lp = pol.forward_and_store_host(np.zeros((3, 62)), rb, [True, False, True])
    assert a.shape == (3,) and a.dtype == np.int64 and tuple(v.shape) == (3, 1) and tuple(lp.shape) == (3,)
    assert hasattr(v, "numpy")                                              # tensors unless as_numpy
    assert set(pol._host_out) == {(3, 62)} and list(pol._host_out[(3, 62)].es) == [1.0, 0.0, 1.0]
    pol.forward_and_store_host(np.zeros((1, 62)), rb, [False], as_numpy=True)
    assert set(pol._host_out) == {(3, 62), (1, 62)} and rb.pos == 2
    with pytest.raises(ValueError):
        pol.forward_and_store_host(np.zeros((1, 61)), rb, [False])
    rb.buffer_size = 2
    with pytest.raises(nat.NativeError):
        pol.forward_and_store_host(np.zeros((1, 62)), rb, [False])


def test_policies_with_another_parameter_layout_are_refused_on_the_fused_mlp_paths():
    """AdapPolicyMult / ModularPolicy carry the plain MlpPolicy `spec` for their rollout buffer but a different parameter vector:
    the entry points that hand `params` to the fused MLP kernels (VecOnPolicyAgent, PPO.train_joint, the exchange / round-robin
    engines behind them) refuse them instead of training the wrong network silently (no GPU: the check runs before any native call)"""
    import pytest

    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.adap import AdapPolicyMult
    from pantheonrl_amd.modular import ModularPolicy
    from pantheonrl_amd.ppo import ActorCriticPolicy, require_mlp_kernels
    assert ActorCriticPolicy.fused_mlp_kernels and not AdapPolicyMult.fused_mlp_kernels and not ModularPolicy.fused_mlp_kernels
    require_mlp_kernels(type("P", (), {})(), "x")                       # a plain policy object passes
    for cls in (AdapPolicyMult, ModularPolicy):
        with pytest.raises(nat.NativeError, match="fused MLP kernels"):
            require_mlp_kernels(cls.__new__(cls), "VecOnPolicyAgent")


def test_policy_kwargs_are_accepted_for_the_default_network_and_refused_by_name_otherwise():
    """trainer.py:108-126,196-203 splat --ego-config / --alt-config JSON into the constructor: `policy_kwargs` is a constructor argument,
    the MlpPolicy default (modular/policies.py:112-114) passes, anything else is a named refusal AT CONSTRUCTION (no GPU needed: the
    check runs before the device is touched)"""
    import pytest

    from pantheonrl_amd import _native as nat
    from pantheonrl_amd.ppo import PPO, UnsupportedPolicyConfig, check_policy_kwargs
    ok = [None, {}, {"net_arch": [dict(pi=[64, 64], vf=[64, 64])]}, {"net_arch": dict(vf=[64, 64], pi=[64, 64])},
          {"activation_fn": "tanh"}, {"activation_fn": th.nn.Tanh, "ortho_init": False}, {"use_sde": False}]
    for kw in ok:
        out = check_policy_kwargs(kw)
        assert out == ({"ortho_init": False} if kw and "ortho_init" in kw else {})
    bad = [({"net_arch": [32, 32]}, "net_arch"), ({"net_arch": [dict(pi=[64, 64], vf=[32])]}, "net_arch"),
           ({"net_arch": [dict(pi=[64, 64, 64], vf=[64, 64])]}, "net_arch"), ({"activation_fn": th.nn.ReLU}, "activation_fn"),
           ({"use_sde": True}, "use_sde"), ({"features_extractor_class": object}, "features_extractor_class"),
           ({"optimizer_kwargs": {"eps": 1e-8}}, "optimizer_kwargs"), ({"num_partners": 2}, "num_partners")]
    for kw, name in bad:
        with pytest.raises(UnsupportedPolicyConfig, match=name):
            check_policy_kwargs(kw)
        with pytest.raises(UnsupportedPolicyConfig, match=name):       # the constructor itself, as `PPO(policy='MlpPolicy', **config)` reaches it
            PPO("MlpPolicy", None, policy_kwargs=kw, device="cuda")
    # a supported configuration gets past the check and fails only for want of a GPU here (no CPU fallback)
    if not th.cuda.is_available():
        with pytest.raises(nat.NativeError, match="no CPU fallback"):
            PPO("MlpPolicy", None, policy_kwargs=ok[2], device="cuda")


def test_action_space_limits_are_stated_at_construction():
    """Box action spaces of ONE dimension and <= 16 components get the DiagGaussian head (general kernels; `clip_actions`,
    util.py:84-99, clips what the environment gets); anything else is refused by name before the device is touched"""
    import pytest

    from pantheonrl_amd import _native as nat, spaces as sp
    from pantheonrl_amd.ppo import PPO, UnsupportedPolicyConfig, check_action_space
    check_action_space(sp.Discrete(6))
    check_action_space(sp.MultiDiscrete([7, 12]))
    check_action_space(sp.Box(-1, 1, (2,)))
    check_action_space(sp.Box(-1, 1, (nat.PH_MAX_BOX_ACT,)))
    assert sp.action_dim(sp.Box(-1, 1, (5,))) == 5
    spec = sp.make_spec(sp.Box(-1, 1, (4,)), sp.Box(-2, 2, (3,)))
    assert (spec.act.kind, spec.act.n) == (nat.PH_SPACE_BOX, 3)
    lay, lay_d = nat.layout_of(spec), nat.layout_of(sp.make_spec(sp.Box(-1, 1, (4,)), sp.MultiDiscrete([1, 1, 1])))
    assert (lay.A, lay.L) == (3, 3) and lay.P == lay.val_b + 1 + 3 == lay_d.P + 3     # log_std[A] behind val_b
    for bad in (sp.Box(-1, 1, (2, 2)), sp.Box(-1, 1, (nat.PH_MAX_BOX_ACT + 1,))):
        env = type("E", (), dict(observation_space=sp.Box(-1, 1, (4,)), action_space=bad, _is_dummy_space_env=True))()
        for make in (lambda: check_action_space(bad), lambda: PPO("MlpPolicy", env, device="cuda")):
            with pytest.raises(UnsupportedPolicyConfig, match="DiagGaussian"):
                make()
    with pytest.raises(UnsupportedPolicyConfig, match=r"util\.py:84-99"):
        check_action_space(sp.MultiBinary(3))
    from pantheonrl_amd.common.util import clip_actions
    pol = type("P", (), dict(action_space=sp.Box(-1, 1, (2,))))()
    assert np.array_equal(clip_actions(np.array([[-3.0, 0.5]], np.float32), pol), [[-1.0, 0.5]])


def test_gaussian_oracle_is_torch_normal_summed_over_dimensions():
    """the DiagGaussian checker against closed forms: log N(a; mu, sigma) and 0.5 + 0.5 log(2 pi) + log sigma per dimension"""
    import torch as th

    from oracle import sb3_oracle as orc
    pol = orc.GaussianMlpPolicyOracle(orc.SpaceSpec("box", dim=4), orc.SpaceSpec("box", dim=3))
    assert pol.flat_params().size == orc.MlpPolicyOracle(orc.SpaceSpec("box", dim=4), orc.SpaceSpec("discrete", nvec=(3,))).flat_params().size + 3
    with th.no_grad():
        pol.log_std.copy_(th.tensor([0.3, -0.7, 0.0]))
    flat = pol.flat_params()
    assert np.array_equal(flat[-3:], np.float32([0.3, -0.7, 0.0]))
    pol.load_flat_params(flat)
    obs = th.randn(5, 4, generator=th.Generator().manual_seed(0))
    eps = th.randn(5, 3, generator=th.Generator().manual_seed(1))
    with th.no_grad():
        a, v, lp = pol.forward(obs, uniforms=eps)
        mu = pol.forward(obs, deterministic=True)[0]
        v2, lp2, ent = pol.evaluate_actions(obs, a)
    sd = pol.log_std.detach().exp()
    np.testing.assert_allclose(a.numpy(), (mu + sd * eps).numpy(), atol=1e-6)
    ref = (-0.5 * eps ** 2 - pol.log_std.detach() - 0.5 * np.log(2 * np.pi)).sum(1)
    np.testing.assert_allclose(lp.numpy(), ref.numpy(), atol=1e-5)
    np.testing.assert_allclose(lp2.numpy(), lp.numpy(), atol=1e-6)
    np.testing.assert_allclose(ent.numpy(), np.full(5, (0.5 + 0.5 * np.log(2 * np.pi)) * 3 + float(pol.log_std.detach().sum())), atol=1e-5)
    assert th.equal(v, v2)
    mb = dict(observations=obs, actions=a, advantages=th.randn(5), old_log_prob=lp - 0.1, old_values=v.flatten(), returns=th.randn(5))
    hp = orc.PPOHyper()
    hp.ent_coef = 0.01
    loss, _ = orc.ppo_minibatch_loss(pol, mb, hp)
    pol.optimizer.zero_grad()
    loss.backward()
    g = pol.flat_grads()
    assert g.size == flat.size and np.all(np.isfinite(g)) and np.abs(g[-3:]).max() > 0


def test_bench_reports_the_rollout_form_that_was_asked_for():
    """bench.py downgrades --rollout scripted to stepwise for shapes outside the one-launch rollout's class: the JSON line says so
    (config.rollout vs config.rollout_requested), not only stderr"""
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert 'rollout_requested = args.rollout' in src and '"rollout_requested": rollout_requested' in src
    assert src.index('rollout_requested = args.rollout') < src.index('args.rollout = "stepwise"')
