"""CPU: bounds the engine's default sizing relies on (doudizhu-rl_amd/engine.py
MAX_LEGAL_PER_TABLE = 512 rows per table), checked with the oracle."""
import ctypes
import os
import re
import subprocess

import numpy as np

FULL = np.array([4] * 13 + [1, 1])
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUND_SRC = os.path.join(REPO, "tools", "max_legal_bound.c")


def test_497_is_the_exact_maximum_exhaustive(oracle, tmp_path):
    """PROOF of the slab stride / STAGE_CAP sizing: tools/max_legal_bound.c enumerates all 153,009,740 count vectors of
    20-card hands with a closed-form lead count (fewer cards / follows cannot have more, see its header).  The closed
    form itself is checked against the oracle's dense scan first."""
    so, exe = str(tmp_path / "bound.so"), str(tmp_path / "bound")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-DMAXLEGAL_NO_MAIN", "-o", so, BOUND_SRC])
    L = ctypes.CDLL(so)
    rng = np.random.default_rng(8)
    deck = np.repeat(np.arange(15), FULL)
    for k in range(1500):
        n = 20 if k % 2 else int(rng.integers(1, 21))
        h = np.bincount(rng.choice(deck, n, replace=False), minlength=15).astype(np.int8)
        assert L.count_leads(h.ctypes.data_as(ctypes.c_void_p)) == len(oracle.legal(h)), h
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", exe, BOUND_SRC])
    out = subprocess.check_output([exe], timeout=600).decode()
    assert int(re.search(r"visited: (\d+)", out).group(1)) == 153009740
    assert int(re.search(r"max legal leads: (\d+)", out).group(1)) == 497


def test_joker_kicker_rule_set_maximum_exhaustive(oracle, tmp_path):
    """The same proof for the optional rule set with the 24 joker-kicker rows (libddz_hip_jk.so: STAGE_CAP, stride and
    MAX_LEGAL_PER_TABLE are 512 there): the closed form built with -DDDZ_JK_RULES equals the joker-kicker oracle's dense scan on
    hands that own the extra rows, and its exhaustive maximum over all 20-card hands fits the 512-row slab."""
    so, exe = str(tmp_path / "bound_jk.so"), str(tmp_path / "bound_jk")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-DDDZ_JK_RULES", "-DMAXLEGAL_NO_MAIN", "-o", so, BOUND_SRC])
    L = ctypes.CDLL(so)
    rng = np.random.default_rng(9)
    deck13 = np.repeat(np.arange(13), 4)
    with oracle.variant(jk=True):
        for k in range(600):
            n = 20 if k % 2 else int(rng.integers(4, 21))
            h = np.zeros(15, np.int8)
            h[13] = h[14] = 1                      # both jokers: the hands the extra rows exist for
            h[:13] = np.bincount(rng.choice(deck13, n - 2, replace=False), minlength=13)
            if k % 3 == 0:
                r = int(rng.integers(0, 11))
                h[r] = max(h[r], 3); h[r + 1] = max(h[r + 1], 3)
                while h.sum() > 20:
                    j = int(rng.choice(np.flatnonzero(h[:13] > 0)))
                    if j not in (r, r + 1):
                        h[j] -= 1
            assert L.count_leads(h.ctypes.data_as(ctypes.c_void_p)) == len(oracle.legal(h)), h
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-DDDZ_JK_RULES", "-o", exe, BOUND_SRC])
    out = subprocess.check_output([exe], timeout=600).decode()
    assert int(re.search(r"visited: (\d+)", out).group(1)) == 153009740
    best = int(re.search(r"max legal leads: (\d+)", out).group(1))
    assert best == 497 <= 512   # the worst hands hold no jokers: the extra rows do not raise the maximum


def test_known_worst_20_card_hand(oracle):
    # 4 consecutive triples inside a 12-card straight: a hand that attains the proven maximum
    worst = np.array([1, 1, 1, 1, 1, 3, 3, 3, 3, 1, 1, 1, 0, 0, 0], np.int8)
    assert worst.sum() == 20
    assert len(oracle.legal(worst)) == 497 < 512


def test_random_search_stays_below_capacity(oracle):
    rng = np.random.default_rng(3)
    deck = np.repeat(np.arange(15), FULL)
    best = 0
    for restart in range(40):
        h = np.bincount(rng.choice(deck, 20, replace=False), minlength=15)
        cur = len(oracle.legal(h))
        for _ in range(250):
            a = rng.choice(np.flatnonzero(h > 0)); b = rng.choice(np.flatnonzero(h < FULL))
            if a == b:
                continue
            h2 = h.copy(); h2[a] -= 1; h2[b] += 1
            c2 = len(oracle.legal(h2))
            if c2 >= cur:
                h, cur = h2, c2
        best = max(best, cur)
    assert 300 < best <= 497


def test_follow_lists_are_small(oracle):
    rows, info = oracle.action_table()
    rng = np.random.default_rng(4)
    deck = np.repeat(np.arange(15), FULL)
    worst = 0
    for _ in range(300):
        h = np.bincount(rng.choice(deck, 20, replace=False), minlength=15)
        last = rows[rng.integers(1, 13527), :15]
        worst = max(worst, len(oracle.legal(h, last)))
    assert worst < 64   # a follow list is pass + same-category beats + bombs + rocket


def test_episode_length_bound(oracle):
    # every non-pass play removes a card and at most two passes separate plays: <= 162 plies,
    # so the u16 ply counter and the RNG's 16-bit ply field cannot wrap
    env = oracle.OracleEnv(64, seed=2)
    env.reset()
    longest = 0
    for _ in range(200):
        env.legal()
        ply = env.field(10)[:, 4:6].copy().view(np.uint16)[:, 0]
        longest = max(longest, int(ply.max()))
        env.step(oracle.STEP_RANDOM, auto_reset=True)
    assert 30 < longest <= 162
