"""not-gpu: the C-ABI library loads, exports every symbol include/pantheon_hip.h declares, its host-only entry points
work, and it fails loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch as th

from pantheonrl_amd import _native as nat
from pantheonrl_amd import spaces as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "pantheon_hip.h")).read()
DECLARED = sorted(set(re.findall(r"^(?:int|const char \*)\s*(ph_[a-z_0-9]+)\s*\(", HEADER, flags=re.M)))


def test_library_exports_every_declared_symbol():
    lib = nat.load()
    assert len(DECLARED) >= 25
    for name in DECLARED:
        assert hasattr(lib, name), f"{name} declared in include/pantheon_hip.h but not exported"
    assert set(DECLARED) == set(nat.SIGNATURES), set(DECLARED) ^ set(nat.SIGNATURES)
    assert lib.ph_abi_version() == 7
    assert int(re.search(r"#define PH_NSTAT (\d+)", HEADER).group(1)) == nat.PH_NSTAT
    assert int(re.search(r"#define PH_MAX_COMP (\d+)", HEADER).group(1)) == nat.PH_MAX_COMP


def test_struct_sizes_match_the_header_layout():
    assert C.sizeof(nat.PhSpace) == 4 * (2 + nat.PH_MAX_COMP)
    assert C.sizeof(nat.PhSpec) == 2 * C.sizeof(nat.PhSpace)
    assert C.sizeof(nat.PhLayout) == 4 * 17
    assert C.sizeof(nat.PhRollout) == 8 + 8 * 8
    assert C.sizeof(nat.PhPpoHyper) == 4 * 11
    assert C.sizeof(nat.PhOptState) == 8 * 4


@pytest.mark.parametrize("obs,act,expect", [
    (sp.Discrete(1), sp.Discrete(3), dict(D=1, F=1, A=1, L=3, P=8836)),                                  # RPS
    (sp.MultiDiscrete([7] * 6 + [7, 12] * 12), sp.MultiDiscrete([7, 12]), dict(D=30, F=270, A=2, L=19, P=44308)),
    (sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6), dict(D=62, F=62, A=1, L=6, P=16839)),               # Overcooked
    (sp.Box(-np.inf, np.inf, (48,)), sp.Discrete(5), dict(D=48, F=48, A=1, L=5, P=14982)),               # MPE N=8
])
def test_layout_matches_survey_parameter_counts(obs, act, expect):
    lay = nat.layout_of(sp.make_spec(obs, act))
    for k, v in expect.items():
        assert getattr(lay, k) == v, k
    assert lay.pi_W1 == 0 and lay.pi_b1 == lay.F * 64 and lay.val_b == lay.P - 1
    assert lay.act_W + 64 * lay.L == lay.act_b and lay.val_W == lay.act_b + lay.L


def test_bad_specs_are_rejected_with_a_message():
    for bad in (sp.Box(-1, 1, (2, 2)), sp.Box(-1, 1, (nat.PH_MAX_BOX_ACT + 1,)), sp.MultiBinary(3)):
        with pytest.raises(sp.SpaceException):
            sp.make_spec(sp.Box(-1, 1, (3,)), bad)
    spec = sp.make_spec(sp.Box(-1, 1, (3,)), sp.Box(-1, 1, (2,)))      # Box actions: the A means + log_std[A] behind val_b
    lay = nat.layout_of(spec)
    assert (lay.A, lay.L, lay.P) == (2, 2, lay.val_b + 1 + 2)
    spec.act.n = nat.PH_MAX_BOX_ACT + 1
    with pytest.raises(nat.NativeError, match="PH_MAX_BOX_ACT"):
        nat.layout_of(spec)
    spec = sp.make_spec(sp.Box(-1, 1, (3,)), sp.Discrete(4))
    spec.act.nvec[0] = 0
    with pytest.raises(nat.NativeError, match="nvec"):
        nat.layout_of(spec)
    spec = sp.make_spec(sp.Box(-1, 1, (3,)), sp.MultiDiscrete([40, 40]))
    with pytest.raises(nat.NativeError, match="logits"):
        nat.layout_of(spec)


def test_feistel_permutation_is_a_bijection_and_keyed():
    for n in (1, 2, 3, 5, 64, 100, 1000, 2048, 4097):
        p = nat.feistel_indices(n, 11, 0)
        assert sorted(p.tolist()) == list(range(n))
        if n >= 64:
            assert (p != nat.feistel_indices(n, 11, 1)).mean() > 0.9      # a fresh permutation per epoch
            assert (p != nat.feistel_indices(n, 12, 0)).mean() > 0.9      # and per seed
            assert np.array_equal(p[10:20], nat.feistel_indices(n, 11, 0, start=10, count=10))
    big = nat.feistel_indices(131072, 3, 2)
    assert len(np.unique(big)) == 131072
    # crude uniformity: each quarter of the output draws evenly from each quarter of the input range
    q = (big.reshape(4, -1) // 32768)
    for row in q:
        counts = np.bincount(row, minlength=4)
        assert (np.abs(counts - 8192) < 400).all()
    with pytest.raises(nat.NativeError):
        nat.feistel_indices(10, 0, 0, start=5, count=6)


@pytest.mark.skipif(th.cuda.is_available(), reason="checks the no-device error path")
def test_no_cpu_fallback():
    lib = nat.load()
    h = C.c_void_p()
    assert lib.ph_ctx_create(0, C.byref(h)) != 0 and not h.value
    assert b"no HIP device" in lib.ph_last_error() or b"fallback" in lib.ph_last_error()
    n = C.c_int(-1)
    lib.ph_device_count(C.byref(n))
    assert n.value == 0
    from pantheonrl_amd import PPO
    env = type("E", (), dict(observation_space=sp.Discrete(1), action_space=sp.Discrete(3)))()
    with pytest.raises(nat.NativeError, match="no CPU fallback"):
        PPO("MlpPolicy", env)
    with pytest.raises(nat.NativeError, match="no CPU fallback"):
        PPO("MlpPolicy", env, device="cpu")
    # null-handle calls report an error instead of crashing
    assert lib.ph_ctx_sync(None) != 0 and lib.ph_gae(None, None, None, None, 0.99, 0.95, 0) != 0


def test_split_kernel_tables_cover_every_parameter_exactly_as_documented():
    """Host-only check of the two tables that tie ppo_grad_split_kernel to the flat parameter vector (ph_debug_split_tables):
    * slab map: a workgroup's gradient slab holds every parameter exactly once (padding elsewhere); position
      ((wave * 4 + blk) * 64 + lane) * 4 + r of the dW2 / dW1 sections is element (k = 16 blk + 4 (lane >> 4) + r,
      col = 16 wave + (lane & 15)); with the folded bias, input row 63 of dW1 is d b1;
    * weight image: [net][wave][set][chunk][plane][lane][8]; W1[k][n] backs set 0 of wave n / 16 at lane ((k / 8) % 4) * 16 + n % 16,
      chunk k / 32, slot k % 8; W2[k][n] backs set 1 the same way AND set 2 of wave k / 16 at lane ((n / 8) % 4) * 16 + k % 16,
      chunk n / 32, slot n % 8; b1[n] rides as feature 63 of set 0 when folded; nothing else is backed."""
    import ctypes as C

    import numpy as np

    from pantheonrl_amd import _native as nat, spaces as sp
    lib = nat.load()
    RS_NET, RS_W2, RS_W1, RS_B1, RS_B2, RS_HW, RS_HB = 8960, 0, 4096, 8192, 8256, 8320, 8832

    def elem(net, wave, st, c, lane, e):
        return (((((net * 4 + wave) * 3 + st) * 2 + c) * 3 + 0) * 64 + lane) * 8 + e

    for F, L in ((62, 6), (64, 8), (1, 2), (48, 5)):
        spec = sp.make_spec(sp.Box(-np.inf, np.inf, (F,)), sp.Discrete(L))
        lay = nat.layout_of(spec)
        slab = (C.c_int * (2 * RS_NET))()
        img = (C.c_int * (2 * lay.P))()
        ok = C.c_int(0)
        nat.check(lib.ph_debug_split_tables(C.byref(spec), slab, img, C.byref(ok)))
        assert ok.value == 1
        slab, img = np.array(slab).reshape(2, RS_NET), np.array(img).reshape(lay.P, 2)
        fold = F < 64
        held = slab[slab >= 0]
        assert sorted(held.tolist()) == list(range(lay.P))                      # every parameter once, nothing twice
        for net, (oW1, oB1, oW2, oB2) in enumerate(((lay.pi_W1, lay.pi_b1, lay.pi_W2, lay.pi_b2),
                                                    (lay.vf_W1, lay.vf_b1, lay.vf_W2, lay.vf_b2))):
            for wave, blk, lane, r in ((0, 0, 0, 0), (3, 3, 63, 3), (1, 2, 37, 1), (2, 3, 48, 3)):
                pos = ((wave * 4 + blk) * 64 + lane) * 4 + r
                k, col = 16 * blk + 4 * (lane >> 4) + r, 16 * wave + (lane & 15)
                assert slab[net, RS_W2 + pos] == oW2 + k * 64 + col
                want = oW1 + k * 64 + col if k < F else (oB1 + col if (fold and k == 63) else -1)
                assert slab[net, RS_W1 + pos] == want
            assert np.array_equal(slab[net, RS_B2:RS_B2 + 64], oB2 + np.arange(64))
            assert np.array_equal(slab[net, RS_B1:RS_B1 + 64], np.full(64, -1) if fold else oB1 + np.arange(64))
            for k, n in ((0, 0), (F - 1, 63), (min(F - 1, 17), 42)):
                by_col = elem(net, n >> 4, 0, k >> 5, ((k >> 3) & 3) * 16 + (n & 15), k & 7)
                assert img[oW1 + k * 64 + n].tolist() == [by_col, -1]
            for k, n in ((0, 0), (63, 63), (21, 40)):
                fwd = elem(net, n >> 4, 1, k >> 5, ((k >> 3) & 3) * 16 + (n & 15), k & 7)
                bwd = elem(net, k >> 4, 2, n >> 5, ((n >> 3) & 3) * 16 + (k & 15), n & 7)
                assert img[oW2 + k * 64 + n].tolist() == [fwd, bwd]
            for n in (0, 31, 63):
                want = [elem(net, n >> 4, 0, 1, 3 * 16 + (n & 15), 7), -1] if fold else [-1, -1]
                assert img[oB1 + n].tolist() == want
            assert (img[oB2:oB2 + 64] == -1).all()
        assert (img[lay.act_W:lay.P] == -1).all()                               # heads are not MFMA operands of this kernel
        backed = img[img >= 0]
        assert len(set(backed.tolist())) == len(backed)                          # no two parameters share an image element
    # a spec outside the kernel's class: one-hot observations
    spec = sp.make_spec(sp.MultiDiscrete([3, 4]), sp.Discrete(3))
    ok = C.c_int(1)
    dummy = (C.c_int * 4)()
    nat.check(lib.ph_debug_split_tables(C.byref(spec), dummy, dummy, C.byref(ok)))
    assert ok.value == 0


def test_one_hot_split_kernel_tables_cover_every_parameter():
    """Host-only check of ppo_grad_split_oh_kernel's two tables (ph_debug_split_oh_tables) on Liar's Dice (F = 270: five chunks,
    19 logits) and on a two-chunk / 20-logit shape: a slab holds every parameter exactly once; dW2 / dW1 positions follow
    ((wave 4 + blk) 64 + lane) 4 + r = element (k = 16 blk + 4 (lane / 16) + r, col = 16 wave + lane % 16), chunk by chunk; head
    positions ((lb 4 + wave) 64 + lane) 4 + r = (unit 16 wave + 4 (lane / 16) + r, logit 16 lb + lane % 16); the image backs W1
    once, W2 and the head twice, the biases never, no element twice, everything inside the image."""
    import ctypes as C

    import numpy as np

    from pantheonrl_amd import _native as nat, spaces as sp
    lib = nat.load()
    for obs, act in ((sp.MultiDiscrete([7] * 6 + [7, 12] * 12), sp.MultiDiscrete([7, 12])),
                     (sp.MultiDiscrete([32, 32, 32, 32]), sp.Discrete(20)), (sp.Discrete(5), sp.Discrete(20))):
        spec = sp.make_spec(obs, act)
        lay = nat.layout_of(spec)
        nch = (lay.F + 63) // 64
        n, ne, ok = C.c_int(0), C.c_int(0), C.c_int(0)
        nat.check(lib.ph_debug_split_oh_tables(C.byref(spec), None, None, C.byref(n), C.byref(ne), C.byref(ok)))
        assert ok.value == 1 and n.value % 8 == 0
        nfrag = 8 * nch + 24
        assert ne.value == 2 * nfrag * 3 * 512
        slab = (C.c_int * n.value)()
        img = (C.c_int * (2 * lay.P))()
        nat.check(lib.ph_debug_split_oh_tables(C.byref(spec), slab, img, C.byref(n), C.byref(ne), C.byref(ok)))
        rsn = n.value // 2
        assert rsn == 4096 * (1 + nch) + 128 + 2048 + 32
        slab, img = np.array(slab).reshape(2, rsn), np.array(img).reshape(lay.P, 2)
        assert sorted(slab[slab >= 0].tolist()) == list(range(lay.P))
        for net, (oW1, oB1, oW2, oB2, oHW, oHB, Lh) in enumerate(((lay.pi_W1, lay.pi_b1, lay.pi_W2, lay.pi_b2, lay.act_W, lay.act_b, lay.L),
                                                                   (lay.vf_W1, lay.vf_b1, lay.vf_W2, lay.vf_b2, lay.val_W, lay.val_b, 1))):
            for wave, blk, lane, r in ((0, 0, 0, 0), (3, 3, 63, 3), (1, 2, 37, 1), (2, 3, 48, 3)):
                pos = ((wave * 4 + blk) * 64 + lane) * 4 + r
                k, col = 16 * blk + 4 * (lane >> 4) + r, 16 * wave + (lane & 15)
                assert slab[net, pos] == oW2 + k * 64 + col
                for ch in range(nch):
                    want = oW1 + (64 * ch + k) * 64 + col if 64 * ch + k < lay.F else -1
                    assert slab[net, 4096 * (1 + ch) + pos] == want
            b1 = 4096 * (1 + nch)
            assert np.array_equal(slab[net, b1:b1 + 64], oB1 + np.arange(64)) and np.array_equal(slab[net, b1 + 64:b1 + 128], oB2 + np.arange(64))
            hw = b1 + 128
            for lb, wave, lane, r in ((0, 0, 0, 0), (1, 3, 63, 3), (0, 2, 21, 1), (1, 1, 3, 2)):
                u, l = 16 * wave + 4 * (lane >> 4) + r, 16 * lb + (lane & 15)
                assert slab[net, hw + ((lb * 4 + wave) * 64 + lane) * 4 + r] == (oHW + u * Lh + l if l < Lh else -1)
            assert np.array_equal(slab[net, hw + 2048:hw + 2048 + Lh], oHB + np.arange(Lh))

            def elem(frag, lane, e):
                return ((net * nfrag + frag) * 3) * 512 + lane * 8 + e
            for f, n_ in ((0, 0), (lay.F - 1, 63), (min(lay.F - 1, 70), 42)):
                k = f & 63
                assert img[oW1 + f * 64 + n_].tolist() == [elem(((n_ >> 4) * nch + (f >> 6)) * 2 + (k >> 5), ((k >> 3) & 3) * 16 + (n_ & 15), k & 7), -1]
            for k, n_ in ((0, 0), (63, 63), (21, 40)):
                fwd = elem(8 * nch + (n_ >> 4) * 2 + (k >> 5), ((k >> 3) & 3) * 16 + (n_ & 15), k & 7)
                bwd = elem(8 * nch + 8 + (k >> 4) * 2 + (n_ >> 5), ((n_ >> 3) & 3) * 16 + (k & 15), n_ & 7)
                assert img[oW2 + k * 64 + n_].tolist() == [fwd, bwd]
            for u, l in ((0, 0), (63, Lh - 1), (37, min(Lh - 1, 17))):
                hz = elem(8 * nch + 16 + (l >> 4) * 2 + (u >> 5), ((u >> 3) & 3) * 16 + (l & 15), u & 7)
                hd = elem(8 * nch + 20 + (u >> 4), ((l >> 3) & 3) * 16 + (u & 15), l & 7)
                assert img[oHW + u * Lh + l].tolist() == [hz, hd]
            for o in (oB1, oB2):
                assert (img[o:o + 64] == -1).all()
            assert (img[oHB:oHB + Lh] == -1).all()
        backed = img[img >= 0]
        assert len(set(backed.tolist())) == len(backed) and backed.max() + 2 * 512 < ne.value
    spec = sp.make_spec(sp.Box(-np.inf, np.inf, (62,)), sp.Discrete(6))   # Box observations: not this kernel's
    ok = C.c_int(1)
    nat.check(lib.ph_debug_split_oh_tables(C.byref(spec), None, None, C.byref(n), C.byref(ne), C.byref(ok)))
    assert ok.value == 0
