#!/usr/bin/env python3
"""Generate tests/golden/rule_agent.npz (fixtures G7 + G8) -- runs ONLY in the build container.

  G7 cards_value      the list rule_based/utils/evaluator.py:10-47 builds at import (float64[13527])
  G8 hand/last/left/role -> choice
                      the action RuleBasedModel.choose (rule_based/utils/rule_based_model.py:43-101) returns, with
                      Decomposer.get_combinations (rule_based/utils/decomposer.py:17-76) underneath: the REFERENCE's
                      own Python, imported from /root/reference and executed here.

What had to be supplied for that import (the fixture is data: inputs + the outputs of the reference's code):
  * module `env` with get_combinations_nosplit / get_combinations_recursive: the native module is absent from the
    reference (precompiled/ is empty), so these two are this repo's "decomposer spec v1" stand-ins
    (oracle/ddz_auto_oracle.c: ddzo_combinations_nosplit / _recursive) -- the part of row N1 that is PARITY UNPINNED.
    Everything above them (valid-row filtering, index mapping, clamp_action_idx, fine_mask, scoring, tie-breaking)
    is the reference's code.
  * module `tensorflow`: rule_based/utils/utils.py:5 imports it at module level but the functions used here
    (get_mask_onehot60) never touch it; an empty module object stands in.
  * numpy >= 1.24 removed the aliases np.int / np.bool the reference still uses (decomposer.py:42,
    rule_based_model.py:37): restored as int / bool for this process.
  * an env object with the five members choose() reads (rule_based_model.py:17-33,56,97-98): hand, last two handouts,
    role id, `left`, and the cards2arr / arr2cards codecs of envi.py:118-137.
"""
import contextlib
import io
import os
import sys
import time
import types

import numpy as np

sys.dont_write_bytecode = True  # never write into /root/reference
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle import oracle  # noqa: E402  (stand-ins for the two absent native functions)

np.int = int    # noqa: removed numpy aliases the reference uses
np.bool = bool  # noqa

_env = types.ModuleType("env")
_env.get_combinations_nosplit = lambda mask, card_mask: oracle.combinations_nosplit(
    np.asarray(mask) != 0, np.asarray(card_mask) != 0)
_env.get_combinations_recursive = lambda mask, target: oracle.combinations_recursive(
    np.asarray(mask, np.uint8), np.asarray(target, np.uint8))
sys.modules["env"] = _env
sys.modules["tensorflow"] = types.ModuleType("tensorflow")

from rule_based.utils.evaluator import cards_value  # noqa: E402  (reference)
from rule_based.utils.rule_based_model import RuleBasedModel  # noqa: E402  (reference)


class PayloadEnv:
    """the members of envi.Env that RuleBasedModel.choose reads"""

    def __init__(self, hand, last, left, role):
        self._hand, self._last, self.left, self._role = hand, last, np.array(left), int(role)

    @staticmethod
    def arr2cards(arr):  # envi.py:118-130
        return np.repeat(np.arange(3, 18), np.asarray(arr, dtype=int))

    @staticmethod
    def cards2arr(cards):  # envi.py:132-137
        arr = np.zeros((15,), dtype=int)
        for c in cards:
            arr[int(c) - 3] += 1
        return arr

    def get_curr_handcards(self):
        return self.arr2cards(self._hand)

    def get_last_two_cards(self):  # [previous player's handout, the one before]; choose() takes the first non-empty
        return [self.arr2cards(self._last).tolist(), []]

    def get_role_ID(self):
        return self._role + 1


def ref_choice(args):
    hand, last, left, role = args
    with contextlib.redirect_stdout(io.StringIO()):  # choose() prints the hand and the move
        onehot = RuleBasedModel().choose(PayloadEnv(hand, last, left, role))
    counts = np.asarray(onehot).reshape(15, 4).sum(axis=1).astype(np.int8)
    a = oracle.lookup(counts)
    assert a >= 0
    return a


def game_states(T, iters, seed, auto_roles):
    env = oracle.OracleEnv(T, seed=seed)
    env.reset()
    out = []
    for _ in range(iters):
        st = env.state.reshape(T, 11, 16)
        for t in range(T):
            if st[t, 10, 1] or not st[t, 10, 6]:
                continue
            role = int(st[t, 10, 0])
            b1, b2 = st[t, 6 + (role + 2) % 3, :15], st[t, 6 + (role + 1) % 3, :15]
            last = (b1 if b1.any() else b2).astype(np.int8)
            out.append((st[t, role, :15].astype(np.int8).copy(), last.copy(), st[t, 0:3, 15].astype(np.int32).copy(), role))
        ids = env.auto_choose(auto_roles)
        env.legal()
        env.step(oracle.STEP_IDS, ids, auto_reset=True)
    return out


def main():
    import multiprocessing as mp
    quick = "--quick" in sys.argv
    t0 = time.time()
    cv = np.array(cards_value, np.float64)
    assert cv.shape == (13527,)
    assert np.array_equal(cv, oracle.cards_value()), "oracle restatement of evaluator.py differs"
    rng = np.random.default_rng(20261006)
    cases = game_states(24, 70, seed=31, auto_roles=0b101) + game_states(12, 60, seed=32, auto_roles=0b111)
    # cap the cost of the Python reference on the heaviest states, keep every hand size and both decomposers
    stats = np.array([oracle.auto_choose(h, l if l.any() else None, f, r, want_stats=True)[1][0] for h, l, f, r in cases])
    keep = [k for k in range(len(cases)) if stats[k] <= 4000]
    pick = rng.choice(keep, min(len(keep), 200 if quick else 2600), replace=False)
    cases = [cases[k] for k in sorted(pick)]
    rows, _ = oracle.action_table()
    deck = np.repeat(np.arange(15), [4] * 13 + [1, 1])
    for k in range(40 if quick else 500):   # synthetic queries: arbitrary last / card counts / role
        m = int(rng.integers(1, 15))
        hand = np.bincount(rng.choice(deck, m, replace=False), minlength=15).astype(np.int8)
        last = rows[int(rng.integers(1, 13527)), :15].copy() if rng.random() < 0.6 else np.zeros(15, np.int8)
        left = rng.integers(1, 21, 3).astype(np.int32)
        c = (hand, last, left, int(rng.integers(0, 3)))
        if oracle.auto_choose(hand, last if last.any() else None, left, c[3], want_stats=True)[1][0] <= 4000:
            cases.append(c)
    print(f"{len(cases)} cases, running the reference's choose() ...", flush=True)
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        choice = pool.map(ref_choice, cases, chunksize=8)
    choice = np.array(choice, np.int32)
    mine = np.array([oracle.auto_choose(h, l if l.any() else None, f, r) for h, l, f, r in cases], np.int32)
    bad = np.flatnonzero(mine != choice)
    print(f"oracle restatement vs reference choose(): {len(bad)} mismatches of {len(cases)} ({time.time() - t0:.0f}s)")
    for k in bad[:10]:
        print("  ", cases[k], "reference", choice[k], "oracle", mine[k])
    assert bad.size == 0
    np.savez_compressed(os.path.join(HERE, "rule_agent.npz"), cards_value=cv,
                        hand=np.stack([c[0] for c in cases]), last=np.stack([c[1] for c in cases]),
                        left=np.stack([c[2] for c in cases]).astype(np.int8),
                        role=np.array([c[3] for c in cases], np.int8), choice=choice)
    follow = np.stack([c[1] for c in cases]).any(1)
    print(f"wrote rule_agent.npz: {len(cases)} cases ({follow.sum()} follows, {(choice[follow] == 0).sum()} passes)")


if __name__ == "__main__":
    main()
