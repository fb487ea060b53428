"""ctypes binding of libpantheon_hip.so (include/pantheon_hip.h).

There is no CPU fallback: if the shared library cannot be loaded, or no gfx950 device is visible when a context is
created, the engine raises.  Loading the library itself needs no GPU (the not-gpu tests check the exported symbols).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

PH_HIDDEN = 64
PH_MAX_COMP = 256
PH_MAX_LOGITS = 64
PH_MAX_BOX_ACT = 16
PH_NSTAT = 8
PH_SPACE_BOX = 0
PH_SPACE_DISCRETE = 1
STAT_NAMES = ("policy_loss", "value_loss", "entropy_loss", "clip_fraction", "approx_kl", "loss", "grad_norm", "applied")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PANTHEON_HIP_LIB") or os.path.join(_HERE, "csrc", "libpantheon_hip.so")


class NativeError(RuntimeError):
    """Raised for any non-zero status from the C ABI (message from ph_last_error)."""


class PhSpace(C.Structure):
    _fields_ = [("kind", C.c_int), ("n", C.c_int), ("nvec", C.c_int * PH_MAX_COMP)]


class PhSpec(C.Structure):
    _fields_ = [("obs", PhSpace), ("act", PhSpace)]


class PhLayout(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("D", "F", "A", "L", "P", "pi_W1", "pi_b1", "pi_W2", "pi_b2", "vf_W1", "vf_b1",
                                        "vf_W2", "vf_b2", "act_W", "act_b", "val_W", "val_b")]


class PhAdapMultLayout(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("Fo", "C", "L", "P", "pi_W1", "pi_b1", "pi_Ws", "pi_bs", "pi_W2", "pi_b2", "vf_W1", "vf_b1",
                                        "vf_Ws", "vf_bs", "vf_W2", "vf_b2", "act_W", "act_b", "val_W", "val_b")]


class PhRollout(C.Structure):
    _fields_ = [("T", C.c_int), ("E", C.c_int)] + [(k, C.c_void_p) for k in (
        "observations", "actions", "rewards", "episode_starts", "values", "log_probs", "advantages", "returns")]


class PhPpoHyper(C.Structure):
    _fields_ = [("learning_rate", C.c_float), ("clip_range", C.c_float), ("clip_range_vf", C.c_float),
                ("ent_coef", C.c_float), ("vf_coef", C.c_float), ("max_grad_norm", C.c_float),
                ("target_kl", C.c_float), ("normalize_advantage", C.c_int), ("adam_beta1", C.c_float),
                ("adam_beta2", C.c_float), ("adam_eps", C.c_float)]


class PhBcLayout(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("D", "F", "A", "L", "P", "W1", "b1", "W2", "b2", "act_W", "act_b", "val_W", "val_b")]


class PhBcHyper(C.Structure):
    _fields_ = [("learning_rate", C.c_float), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float),
                ("ent_weight", C.c_float), ("l2_weight", C.c_float)]


PH_BC_HIDDEN = 32
PH_BC_NSTAT = 8


class PhAdapLoss(C.Structure):
    """ph_adap_loss: ADAP's context term (adap_learn.py:111-116, adap/util.py:97-131)"""
    _fields_ = [("context_size", C.c_int), ("num_context_samples", C.c_int), ("num_state_samples", C.c_int),
                ("sampler", C.c_int), ("context_loss_coeff", C.c_float), ("state_idx", C.c_void_p), ("contexts", C.c_void_p),
                ("seed", C.c_ulonglong), ("context_loss", C.c_void_p), ("used_state_idx", C.c_void_p),
                ("used_contexts", C.c_void_p)]


CONTEXT_SAMPLERS = {"l2": 0, "unit_square": 1, "positive_square": 2, "categorical": 3, "natural_numbers": 4}
BC_STAT_NAMES = ("neglogp", "entropy", "ent_loss", "prob_true_act", "l2_norm", "l2_loss", "loss", "rows")


class PhStepCall(C.Structure):
    _fields_ = [("spec", C.POINTER(PhSpec)), ("params", C.c_void_p), ("obs", C.c_void_p), ("n", C.c_int),
                ("action_mask", C.c_void_p), ("seed", C.c_ulonglong), ("counter", C.c_ulonglong),
                ("deterministic", C.c_int), ("actions_i32", C.c_void_p), ("values", C.c_void_p),
                ("log_probs", C.c_void_p), ("rb", C.POINTER(PhRollout)), ("pos", C.c_int),
                ("episode_start_in", C.c_void_p), ("pending_reward", C.c_void_p), ("joint_actions", C.c_void_p),
                ("n_seats", C.c_int), ("seat", C.c_int), ("partner_seat", C.c_void_p), ("bonus", C.c_float)]


class PhOptState(C.Structure):
    _fields_ = [("params", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("step", C.c_void_p)]


_vp, _i, _ull, _d = C.c_void_p, C.c_int, C.c_ulonglong, C.c_double
# name -> argtypes (restype is int except where noted); the single source the symbol test walks
class PhTrainCall(C.Structure):
    """ph_train_call: one learner's PPO.train() arguments for ph_ppo_train_multi"""
    _fields_ = [("ctx", C.c_void_p), ("spec", C.POINTER(PhSpec)), ("opt", C.POINTER(PhOptState)),
                ("rb", C.POINTER(PhRollout)), ("hyper", C.POINTER(PhPpoHyper)), ("n_epochs", C.c_int),
                ("batch_size", C.c_int), ("perms", C.c_void_p), ("perm_seed", C.c_ulonglong), ("stats", C.c_void_p),
                ("gemm_mode", C.c_int)]


PH_MAX_RANKS = 16


class PhP2P(C.Structure):
    """ph_p2p: the peer-to-peer exchange as seen from one rank"""
    _fields_ = [("world", C.c_int), ("rank", C.c_int), ("count", C.c_int), ("T", C.c_int),
                ("joint", (C.c_void_p * PH_MAX_RANKS) * 2), ("flags", C.c_void_p * PH_MAX_RANKS), ("ll", C.c_void_p * PH_MAX_RANKS), ("ll_slots", C.c_int),
                ("epoch", C.c_void_p), ("error", C.c_void_p), ("timeout_cycles", C.c_ulonglong)]


PH_MOD_MAX = 8


class PhModular(C.Structure):
    """ph_modular: partners and modules of a ModularPolicy"""
    _fields_ = [("num_partners", C.c_int), ("n_modules", C.c_int), ("module_of", C.c_int * PH_MOD_MAX), ("nomain", C.c_int)]


PH_STEP_FIX_ILLEGAL = 2
PH_STEP_MASK_ENV_ONLY = 4


class PhRolloutCall(C.Structure):
    """ph_rollout_call: one local agent's whole scripted rollout in the symmetric exchange layout"""
    _fields_ = [("spec", C.POINTER(PhSpec)), ("params", C.c_void_p), ("obs_seq", C.c_void_p), ("rew_seq", C.c_void_p),
                ("done_seq", C.c_void_p), ("mask_seq", C.c_void_p), ("n", C.c_int), ("episode_start0", C.c_void_p),
                ("seed", C.c_ulonglong), ("counter0", C.c_ulonglong), ("mask_mode", C.c_int), ("actions_i32", C.c_void_p),
                ("values", C.c_void_p), ("log_probs", C.c_void_p), ("rb", C.POINTER(PhRollout)), ("n_seats", C.c_int),
                ("seat", C.c_int), ("partner_seat", C.c_void_p), ("bonus", C.c_float)]


class PhRRLink(C.Structure):
    """ph_rr_link: the engine-side round-robin layout's peer-mapped receive areas"""
    _fields_ = [("n_partners", C.c_int), ("rank", C.c_int), ("n", C.c_int), ("block_ld", C.c_int),
                ("area", C.c_void_p * PH_MAX_RANKS), ("error", C.c_void_p), ("timeout_cycles", C.c_ulonglong)]


class PhRREgo(C.Structure):
    _fields_ = [("spec", C.POINTER(PhSpec)), ("params", C.c_void_p), ("obs_seq", C.c_void_p), ("base_reward_seq", C.c_void_p),
                ("done_seq", C.c_void_p), ("blocks", C.c_void_p), ("partnerid", C.c_void_p), ("rewards", C.c_void_p),
                ("alt_actions", C.c_void_p), ("partner_trace", C.c_void_p), ("episode_start0", C.c_void_p),
                ("seed", C.c_ulonglong), ("counter0", C.c_ulonglong), ("values", C.c_void_p), ("log_probs", C.c_void_p),
                ("rb", C.POINTER(PhRollout)), ("bonus", C.c_float)]


class PhRRPartner(C.Structure):
    _fields_ = [("spec", C.POINTER(PhSpec)), ("params", C.c_void_p), ("obs_scratch", C.c_void_p), ("es_scratch", C.c_void_p),
                ("can_scratch", C.c_void_p), ("pos", C.c_void_p), ("boundary", C.c_void_p), ("term", C.c_void_p),
                ("open", C.c_void_p), ("prev_mask", C.c_void_p), ("seed", C.c_ulonglong), ("counter0", C.c_ulonglong),
                ("actions", C.c_void_p), ("values", C.c_void_p), ("log_probs", C.c_void_p), ("rb", C.POINTER(PhRollout))]


class PhLiarSelfPlay(C.Structure):
    """ph_liar_selfplay: every device pointer of the vectorised Liar's Dice self-play step"""
    _fields_ = [("n", C.c_int), ("spec", C.POINTER(PhSpec)),
                ("hands", C.c_void_p), ("history", C.c_void_p), ("nmoves", C.c_void_p), ("ego_first", C.c_void_p),
                ("dice_seed", C.c_ulonglong), ("probegostart", C.c_float),
                ("ego_params", C.c_void_p), ("ego_rb", C.POINTER(PhRollout)), ("ego_actions", C.c_void_p),
                ("ego_values", C.c_void_p), ("ego_log_probs", C.c_void_p), ("ego_episode_start", C.c_void_p),
                ("ego_seed", C.c_ulonglong),
                ("alt_params", C.c_void_p), ("alt_rb", C.POINTER(PhRollout)), ("alt_actions", C.c_void_p),
                ("alt_values", C.c_void_p), ("alt_log_probs", C.c_void_p), ("alt_pos", C.c_void_p),
                ("alt_boundary", C.c_void_p), ("alt_term", C.c_void_p), ("alt_open", C.c_void_p), ("alt_acted", C.c_void_p),
                ("alt_seed", C.c_ulonglong),
                ("obs_ego", C.c_void_p), ("obs_alt", C.c_void_p), ("episodes", C.c_void_p),
                ("obs_next", C.c_void_p), ("rew1", C.c_void_p), ("rew2", C.c_void_p), ("es_alt", C.c_void_p),
                ("done1", C.c_void_p), ("done2", C.c_void_p), ("running", C.c_void_p), ("can", C.c_void_p),
                ("alt_opens", C.c_void_p), ("ego_opens", C.c_void_p), ("done", C.c_void_p),
                ("zeros8", C.c_void_p), ("ones8", C.c_void_p)]


SIGNATURES = {
    "ph_abi_version": [],
    "ph_roundrobin_env_step": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.c_float, _i],
    "ph_last_error": [],
    "ph_device_count": [C.POINTER(_i)],
    "ph_ctx_create": [_i, C.POINTER(_vp)],
    "ph_ctx_destroy": [_vp],
    "ph_ctx_set_stream": [_vp, _vp],
    "ph_ctx_sync": [_vp],
    "ph_graph_begin": [_vp],
    "ph_graph_end": [_vp, C.POINTER(_i)],
    "ph_graph_launch": [_vp, _i],
    "ph_ctx_set_rng_epoch": [_vp, _vp],
    "ph_rng_epoch_advance": [_vp],
    "ph_debug_set_profile_buffer": [_vp, _vp],
    "ph_set_exclusive_device": [_vp, _i],
    "ph_ctx_step_errors": [_vp, C.POINTER(C.c_uint)],
    "ph_debug_weight_image_mismatches": [_vp, C.POINTER(PhSpec), _vp, C.POINTER(C.c_int)],
    "ph_timer_start": [_vp],
    "ph_timer_stop": [_vp, C.POINTER(C.c_float)],
    "ph_layout_of": [C.POINTER(PhSpec), C.POINTER(PhLayout)],
    "ph_debug_split_tables": [C.POINTER(PhSpec), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "ph_debug_split_oh_tables": [C.POINTER(PhSpec), _vp, _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "ph_buffer_add": [_vp, C.POINTER(PhSpec), C.POINTER(PhRollout), _i, _vp, _vp, _vp, _vp, _vp],
    "ph_buffer_add_reward": [_vp, C.POINTER(PhRollout), _i, _vp, _vp],
    "ph_buffer_add_reward_joint": [_vp, C.POINTER(PhRollout), _i, _vp, _vp, _i, _i, _vp, C.c_float],
    "ph_buffer_reset": [_vp, C.POINTER(PhSpec), C.POINTER(PhRollout)],
    "ph_gae": [_vp, C.POINTER(PhRollout), _vp, _vp, _d, _d, _i],
    "ph_policy_forward": [_vp, C.POINTER(PhSpec), _vp, _vp, _i, _vp, _vp, _vp, _ull, _ull, _i, _vp, _vp, _vp, _vp, _vp,
                          _vp, C.POINTER(PhRollout), _i, _vp, _vp, _i],
    "ph_ctx_set_joint_reward_rule": [_vp, _i],
    "ph_adapmult_layout_of": [C.POINTER(PhSpec), _i, _vp],
    "ph_adapmult_forward": [_vp, C.POINTER(PhSpec), _i, _vp, _vp, _i, _vp, _vp, _vp, _ull, _ull, _i, _vp, _vp, _vp, _vp, _vp,
                            _vp, C.POINTER(PhRollout), _i, _vp],
    "ph_adapmult_minibatch_grad": [_vp, C.POINTER(PhSpec), _i, _vp, C.POINTER(PhRollout), C.POINTER(PhPpoHyper), _vp, _i, _vp, _vp,
                                   _vp],
    "ph_adapmult_train": [_vp, C.POINTER(PhSpec), _i, C.POINTER(PhOptState), C.POINTER(PhRollout), C.POINTER(PhPpoHyper), _i, _i,
                          _vp, _ull, _vp, _vp],
    "ph_policy_act_host": [_vp, C.POINTER(PhSpec), _vp, _vp, _i, _vp, _ull, _ull, _i, _vp, _vp, _vp, C.POINTER(PhRollout), _i, _i],
    "ph_buffer_add_reward_const": [_vp, C.POINTER(PhRollout), _i, C.c_float],
    "ph_policy_step_multi": [_vp, _i, C.POINTER(PhStepCall)],
    "ph_policy_forward_ragged": [_vp, C.POINTER(PhSpec), _vp, _vp, _vp, _ull, _ull, _i, _vp, _vp, _vp, C.POINTER(PhRollout),
                                 _vp, _vp, _vp],
    "ph_buffer_add_reward_ragged": [_vp, C.POINTER(PhRollout), _vp, _vp, _vp],
    "ph_ragged_advance": [_vp, C.POINTER(PhRollout), _vp, _vp],
    "ph_buffer_compact_columns": [_vp, C.POINTER(PhSpec), C.POINTER(PhRollout), C.POINTER(PhRollout), _vp, _i],
    "ph_fix_illegal_actions": [_vp, _vp, _vp, _i, _i],
    "ph_rps_step": [_vp, _vp, _vp, _vp, _vp, _i],
    "ph_liar_step": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i],
    "ph_liar_reset": [_vp, _vp, _vp, _vp, _vp, _vp, _ull, _ull, C.c_float, _i],
    "ph_liar_obs": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i],
    "ph_liar_selfplay_step": [_vp, C.POINTER(PhLiarSelfPlay), _i, _ull, _i],
    "ph_liar_selfplay_rollout": [_vp, C.POINTER(PhLiarSelfPlay), _i, _i, _ull],
    "ph_framestack_push": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "ph_scripted_rollout": [_vp, C.POINTER(PhSpec), _vp, _vp, _vp, _vp, _i, _i, _vp, _ull, _ull, _vp, _vp, _vp,
                            C.POINTER(PhRollout), _i, _i],
    "ph_ppo_train": [_vp, C.POINTER(PhSpec), C.POINTER(PhOptState), C.POINTER(PhRollout), C.POINTER(PhPpoHyper), _i, _i,
                     _vp, _ull, _vp, _i],
    "ph_ppo_train_multi": [C.POINTER(PhTrainCall), _i],
    "ph_comm_unique_id": [_vp],
    "ph_comm_init": [_vp, _vp, _i, _i],
    "ph_comm_destroy": [_vp],
    "ph_all_gather_i32": [_vp, _vp, _vp, _i],
    "ph_selfplay_rollout": [_vp, _i, C.POINTER(PhStepCall), _i, _vp, _vp, _i],
    "ph_p2p_alloc": [_vp, C.c_size_t, C.POINTER(C.c_void_p), _vp],
    "ph_p2p_open": [_vp, _vp, C.POINTER(C.c_void_p)],
    "ph_p2p_close": [_vp, _vp],
    "ph_p2p_free": [_vp, _vp],
    "ph_p2p_push": [_vp, C.POINTER(PhP2P), _vp, _i],
    "ph_p2p_wait": [_vp, C.POINTER(PhP2P), _i],
    "ph_p2p_ll_push": [_vp, C.POINTER(PhP2P), _vp, _i],
    "ph_p2p_ll_unpack": [_vp, C.POINTER(PhP2P), _i],
    "ph_selfplay_rollout_p2p": [_vp, _i, C.POINTER(PhStepCall), _i, _vp, C.POINTER(PhP2P)],
    "ph_selfplay_rollout_persistent": [_vp, _i, C.POINTER(PhRolloutCall), _i, C.POINTER(PhP2P), _i],
    "ph_selfplay_rollout_persistent_capacity": [_vp, C.POINTER(_i)],
    "ph_rr_area_bytes": [_i, _i, _i, C.POINTER(C.c_size_t)],
    "ph_roundrobin_ego_iteration": [_vp, C.POINTER(PhRRLink), C.POINTER(PhRREgo), _i, _ull],
    "ph_roundrobin_partner_iteration": [_vp, C.POINTER(PhRRLink), C.POINTER(PhRRPartner), _i, _ull],
    "ph_modular_layout": [C.POINTER(PhSpec), C.POINTER(PhModular), C.POINTER(PhLayout), C.POINTER(PhLayout), C.POINTER(_i)],
    "ph_modular_forward": [_vp, C.POINTER(PhSpec), C.POINTER(PhModular), _vp, _i, _vp, _i, _vp, _vp, _vp, _ull, _ull, _i, _vp,
                           _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(PhRollout), _i, _vp, _vp, _i],
    "ph_modular_train": [_vp, C.POINTER(PhSpec), C.POINTER(PhModular), C.POINTER(PhOptState), _vp, C.POINTER(PhRollout),
                         C.POINTER(PhPpoHyper), _i, _i, _vp, _ull, _vp, C.c_float, _i],
    "ph_modular_minibatch_grad": [_vp, C.POINTER(PhSpec), C.POINTER(PhModular), _vp, _i, C.POINTER(PhRollout),
                                  C.POINTER(PhPpoHyper), _vp, _i, C.c_float, _vp, _vp, _i],
    "ph_ppo_minibatch_grad": [_vp, C.POINTER(PhSpec), _vp, C.POINTER(PhRollout), C.POINTER(PhPpoHyper), _vp, _i, _vp,
                              _vp, _i],
    "ph_adap_train": [_vp, C.POINTER(PhSpec), C.POINTER(PhOptState), C.POINTER(PhRollout), C.POINTER(PhPpoHyper), _i, _i,
                      _vp, _ull, _vp, _i, C.POINTER(PhAdapLoss)],
    "ph_adap_minibatch_grad": [_vp, C.POINTER(PhSpec), _vp, C.POINTER(PhRollout), C.POINTER(PhPpoHyper), _vp, _i, _vp,
                               _vp, _i, C.POINTER(PhAdapLoss)],
    "ph_bench_ppo_grad": [_vp, C.POINTER(PhSpec), _vp, C.POINTER(PhRollout), C.POINTER(PhPpoHyper), _i, _i, _i,
                          C.POINTER(C.c_float)],
    "ph_bench_gae": [_vp, C.POINTER(PhRollout), _vp, _vp, _d, _d, _i, _i, C.POINTER(C.c_float)],
    "ph_bench_train_kernels": [_vp, C.POINTER(PhSpec), C.POINTER(PhOptState), C.POINTER(PhRollout), C.POINTER(PhPpoHyper),
                               _i, _i, _i, _i, C.POINTER(C.c_float)],
    "ph_feistel_indices": [_i, _ull, _i, _i, _i, C.POINTER(_i)],
    "ph_bc_layout_of": [C.POINTER(PhSpec), C.POINTER(PhBcLayout)],
    "ph_bc_forward": [_vp, C.POINTER(PhSpec), _vp, _vp, _i, _vp, _vp, _vp, _ull, _ull, _i, _vp, _vp, _vp, _vp, _vp],
    "ph_bc_train": [_vp, C.POINTER(PhSpec), C.POINTER(PhOptState), _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(PhBcHyper), _vp],
    # owning-handle layer (host arrays in and out; see include/pantheon_hip.h)
    "ph_agent_last_error": [],
    "ph_agent_create": [_i, C.POINTER(PhSpec), _i, _i, _d, _d, _ull, C.POINTER(_vp)],
    "ph_agent_destroy": [_vp],
    "ph_agent_layout": [_vp, C.POINTER(PhLayout)],
    "ph_agent_set_params": [_vp, _vp],
    "ph_agent_get_params": [_vp, _vp],
    "ph_agent_set_optimizer": [_vp, _vp, _vp, _i],
    "ph_agent_get_optimizer": [_vp, _vp, _vp, C.POINTER(_i)],
    "ph_agent_buffer_reset": [_vp],
    "ph_agent_pos": [_vp, C.POINTER(_i)],
    "ph_agent_act": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "ph_agent_add_reward": [_vp, _vp, _vp],
    "ph_agent_gae": [_vp, _vp, _vp, _i],
    "ph_agent_train": [_vp, C.POINTER(PhPpoHyper), _i, _i, _vp, _ull, _vp],
    "ph_agent_export_buffer": [_vp] + [_vp] * 8,
    "ph_agent_import_buffer": [_vp] + [_vp] * 8 + [_i],
}

_lib: Optional[C.CDLL] = None


def load(build_if_missing: bool = True) -> C.CDLL:
    """dlopen the engine; builds it in-tree with hipcc when the .so is absent or stale and hipcc is available."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and not os.environ.get("PANTHEON_HIP_LIB"):
        try:
            from .csrc import build as _build
            if _build.needs_build():
                _build.build(verbose=False)
        except Exception as exc:  # noqa: BLE001 -- reported below if the library is really unusable
            if not os.path.exists(LIB_PATH):
                raise NativeError(f"libpantheon_hip.so is missing and could not be built: {exc}") from exc
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} not found: build it with `python -m pantheonrl_amd.csrc.build` "
                          "(the MI355X engine has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name in ("ph_last_error", "ph_agent_last_error") else C.c_int
    if lib.ph_abi_version() != 7:
        raise NativeError("libpantheon_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        msg = load().ph_last_error()
        raise NativeError(msg.decode() if msg else f"pantheon_hip error {status}")


def make_space(kind: int, n: int, nvec: Sequence[int] = ()) -> PhSpace:
    s = PhSpace()
    s.kind, s.n = int(kind), int(n)
    if len(nvec) > PH_MAX_COMP:
        raise NativeError("too many MultiDiscrete components")
    for i, v in enumerate(nvec):
        s.nvec[i] = int(v)
    return s


def layout_of(spec: PhSpec) -> PhLayout:
    lay = PhLayout()
    check(load().ph_layout_of(C.byref(spec), C.byref(lay)))
    return lay


def adapmult_layout_of(spec: PhSpec, context_size: int) -> PhAdapMultLayout:
    lay = PhAdapMultLayout()
    check(load().ph_adapmult_layout_of(C.byref(spec), int(context_size), C.byref(lay)))
    return lay


def ptr(t) -> Optional[int]:
    """device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


class Context:
    """Owns one ph_ctx (workspace + stream binding) on one device."""

    def __init__(self, device_index: int = 0):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.ph_ctx_create(int(device_index), C.byref(h)))
        self.handle = h
        self.device_index = int(device_index)
        self.exclusive_hint = False

    def set_stream(self, raw_stream: int) -> None:
        # every engine call binds the caller's current stream first; the native call is skipped while the stream stays the same
        # (a per-step host loop otherwise pays it twice per environment step)
        if raw_stream != getattr(self, "_bound_stream", None):
            check(self.lib.ph_ctx_set_stream(self.handle, C.c_void_p(raw_stream)))
            self._bound_stream = raw_stream

    def sync(self) -> None:
        check(self.lib.ph_ctx_sync(self.handle))

    def set_joint_reward_rule(self, rule: str) -> None:
        """how a step's joint action enters the agents' rewards (ph_ctx_set_joint_reward_rule): "match" = bonus * [own == partner's]
        (the synthetic driver), "rps" = bonus * rock-paper-scissors payoff (rps.py:41-45)"""
        check(self.lib.ph_ctx_set_joint_reward_rule(self.handle, {"match": 0, "rps": 1}[rule]))

    def step_errors(self) -> int:
        """expired waits of the one-launch optimizer step (ph_ctx_step_errors); synchronises the context's stream"""
        n = C.c_uint(0)
        check(self.lib.ph_ctx_step_errors(self.handle, C.byref(n)))
        return int(n.value)

    def set_exclusive_device(self, exclusive: bool) -> None:
        """scheduling hint (ph_set_exclusive_device): this context's training launches have the device to themselves"""
        check(self.lib.ph_set_exclusive_device(self.handle, int(bool(exclusive))))
        self.exclusive_hint = bool(exclusive)

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.ph_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def feistel_indices(n: int, perm_seed: int, epoch: int, start: int = 0, count: Optional[int] = None):
    """host evaluation of the in-kernel minibatch permutation (bit-exact integer statement of the order)."""
    import numpy as np
    count = n - start if count is None else count
    out = (C.c_int * count)()
    check(load().ph_feistel_indices(int(n), int(perm_seed), int(epoch), int(start), int(count), out))
    return np.frombuffer(out, dtype=np.int32).copy()
