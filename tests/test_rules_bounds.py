"""CPU: bounds the engine's default sizing relies on (doudizhu-rl_amd/engine.py
MAX_LEGAL_PER_TABLE = 512 rows per table), checked with the oracle."""
import numpy as np

FULL = np.array([4] * 13 + [1, 1])


def test_known_worst_20_card_hand(oracle):
    # 4 consecutive triples inside a 12-card straight: the maximum found by simulated
    # annealing over 20-card hands (400 restarts x 1500 moves, always the same optimum)
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
