"""CPU: the oracle's rule-based opponent (oracle/ddz_auto_oracle.c) against fixtures G7 / G8, which hold the outputs of
the REFERENCE's own Python (rule_based/utils/evaluator.py, decomposer.py, rule_based_model.py) run in the build
container by tests/golden/gen_rule_agent.py, plus self-consistency of "decomposer spec v1" (the stand-ins for the two
absent native functions; PARITY UNPINNED)."""
import itertools

import numpy as np
import pytest


@pytest.fixture(scope="module")
def g(golden):
    return golden("rule_agent.npz")


def test_g7_cards_value(oracle, g):
    """evaluator.py:10-47 restated == the list the reference builds at import; all values are multiples of 0.5."""
    cv = oracle.cards_value()
    assert np.array_equal(cv, g["cards_value"])
    assert np.array_equal(cv * 2, np.round(cv * 2))
    assert cv[0] == 0 and cv[11498] == 12 and cv[42] == 9          # pass, rocket, bomb
    assert cv[1] == -7 and cv[13] == 5 and cv[16 + 12] == 7.5       # single 3, single 2, pair of 2s (+50 %)


def test_g8_choose_matches_reference(oracle, g):
    """2,850 states: the oracle's choose == RuleBasedModel.choose of the reference (over the spec-v1 stand-ins)."""
    n = len(g["choice"])
    assert n >= 2000
    for k in range(n):
        last = g["last"][k] if g["last"][k].any() else None
        a = oracle.auto_choose(g["hand"][k], last, g["left"][k].astype(np.int32), int(g["role"][k]))
        assert a == g["choice"][k], (k, g["hand"][k], g["last"][k], g["left"][k], g["role"][k], a, g["choice"][k])
    follow = g["last"].any(1)
    assert follow.sum() > 1000 and (g["choice"][follow] == 0).sum() > 300 and (g["choice"][follow] > 0).sum() > 300
    ncards = g["hand"].sum(1)
    assert (ncards > 10).sum() > 500 and (ncards <= 10).sum() > 500


def test_g8_heavy_choose_matches_reference(oracle, golden):
    """G8h: 416 game states above G8's 4,000-combination cap (up to 44,776), each run through the reference's choose()."""
    h = golden("rule_agent_heavy.npz")
    assert len(h["choice"]) >= 400 and h["combinations"].min() > 4000 and h["combinations"].max() > 40000
    for k in range(len(h["choice"])):
        last = h["last"][k] if h["last"][k].any() else None
        a, st = oracle.auto_choose(h["hand"][k], last, h["left"][k].astype(np.int32), int(h["role"][k]), want_stats=True)
        assert a == h["choice"][k] and st[0] == h["combinations"][k] and st[1] == h["nodes"][k], k


def _brute_multisets(rows, target):
    """every multiset of rows summing to target, by brute force over multiplicities (tiny inputs only)"""
    out = set()
    caps = [min((target[k] // r[k]) for k in range(15) if r[k]) for r in rows]
    for mult in itertools.product(*[range(c + 1) for c in caps]):
        tot = sum(m * r for m, r in zip(mult, rows))
        if np.array_equal(tot, target):
            out.add(tuple(i for i, m in enumerate(mult) for _ in range(m)))
    return out


def test_recursive_stand_in_is_every_multiset_once(oracle):
    rng = np.random.default_rng(1)
    rows_all, _ = oracle.action_table()
    for _ in range(25):
        hand = np.bincount(rng.choice(np.repeat(np.arange(6), 3), int(rng.integers(2, 8)), replace=False), minlength=15)
        fit = [r[:15].astype(np.int64) for r in rows_all[1:] if (r[:15] <= hand).all()]
        combs = oracle.combinations_recursive(np.array(fit, np.uint8), hand.astype(np.uint8))
        got = [tuple(sorted(c)) for c in combs]
        assert len(got) == len(set(got))                       # each multiset once
        assert set(got) == _brute_multisets(fit, hand)         # and all of them
        for c in combs:  # depth-first, lowest remaining rank first, ascending row index while the rank stays
            rem = hand.copy()
            prev_rank, prev = -1, -1
            for i in c:
                r = int(np.flatnonzero(rem)[0])
                assert fit[i][r] > 0 and (r != prev_rank or i >= prev)
                rem -= fit[i]
                prev_rank, prev = r, i
            assert not rem.any()
        assert combs == sorted(combs)                          # lexicographic = depth-first order


def test_nosplit_stand_in_is_exact_cover_of_slots(oracle):
    """every combination covers each thermometer slot of the hand exactly once; regular actions never share a rank
    (the "nosplit" property, decomposer.py:32); the list is complete for a hand small enough to brute-force."""
    hand = np.zeros(15, np.int8)
    hand[[0, 1, 2, 3, 4, 5, 6, 7]] = [3, 3, 2, 1, 1, 1, 1, 1]    # 333444 55 6 7 8 9 10: 13 cards -> the nosplit path
    combs = oracle.auto_combinations(hand, False)
    rows, _ = oracle.action_table()
    seen = set()
    for c in combs:
        assert np.array_equal(rows[c][:, :15].sum(0), hand)     # sums to the hand
        key = tuple(c)
        assert key not in seen
        seen.add(key)
    # solo singles/pairs come from the augmented rows: a rank is "owned" by the first action that contains it
    # and every later action containing it must be a plain single or pair of that rank
    for c in combs:
        owner = {}
        for a in c:
            for r in np.flatnonzero(rows[a][:15]):
                if r in owner:
                    assert rows[a][15] in (1, 2) and rows[a][:15].sum() == rows[a][r], (c, a)
                else:
                    owner[r] = a
    # 333444 as plane + kickers 5,6 / triples + ... : spot checks of members and non-members
    ids = {tuple(sorted(c)) for c in combs}
    plane = oracle.lookup(np.array([3, 3, 0, 0, 0, 0, 0, 0] + [0] * 7, np.int8))
    assert any(plane in c for c in ids)
    # "known issue" of the reference (decomposer.py:32): 3334 + 44455 would split rank 4 between two regular actions
    a3334 = oracle.lookup(np.array([3, 1] + [0] * 13, np.int8))
    assert not any(a3334 in c and any(rows[b][1] >= 2 and rows[b][15] not in (1, 2) for b in c if b != a3334) for c in ids)


def test_choose_properties(oracle):
    rows, _ = oracle.action_table()
    # the whole hand in one move is always taken (rule_based_model.py:78-81)
    for a in (5, 20, 60, 400, 430, 500, 700, 9000, 11498, 11600, 12700):
        hand = rows[a][:15]
        assert oracle.auto_choose(hand, None, np.array([9, 9, 9], np.int32), 1) == a
    # nothing beats the rocket: pass, whatever min_oppo_cards says
    rocket = rows[11498][:15]
    hand = rows[42][:15] + rows[7][:15]
    for left in ([2, 2, 2], [17, 20, 17]):
        assert oracle.auto_choose(hand, rocket, np.array(left, np.int32), 0) == 0
    # the chosen move is legal
    rng = np.random.default_rng(4)
    deck = np.repeat(np.arange(15), [4] * 13 + [1, 1])
    for _ in range(300):
        hand = np.bincount(rng.choice(deck, int(rng.integers(1, 18)), replace=False), minlength=15).astype(np.int8)
        last = rows[int(rng.integers(1, 13527)), :15] if rng.random() < 0.5 else None
        a = oracle.auto_choose(hand, last, rng.integers(1, 21, 3).astype(np.int32), int(rng.integers(0, 3)))
        assert a in oracle.legal(hand, last).tolist()


def test_step_ids_mode_and_auto_game(oracle):
    """DDZO_STEP_IDS: ids are validated against the legal list, -1 = engine RNG; rule farmers beat a random lord."""
    T = 200
    env = oracle.OracleEnv(T, seed=2)
    ref = oracle.OracleEnv(T, seed=2)
    env.reset(); ref.reset()
    env.legal(); ref.legal()
    d1 = env.step(oracle.STEP_IDS, np.full(T, -1, np.int32))
    d2 = ref.step(oracle.STEP_RANDOM)
    assert np.array_equal(env.state, ref.state) and all(np.array_equal(a, b) for a, b in zip(d1[:3], d2[:3]))
    lord = farmers = 0
    for _ in range(120):
        ids = env.auto_choose(0b101)
        env.legal()
        done, rew, ill, _ = env.step(oracle.STEP_IDS, ids)
        assert not ill.any()
        lord += int((rew == -1).sum()); farmers += int((rew == 1).sum())
    assert farmers > 2 * lord > 0
