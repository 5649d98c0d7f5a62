#!/usr/bin/env python3
"""Generate tests/golden/rule_agent_heavy.npz (fixture G8h) -- runs ONLY in the build container (about a second of Python per state: the
enumeration itself is the oracle's C stand-in, the scoring loop over its combinations is the reference's Python).

The decisions G8 leaves out: game states whose hand decomposes into MORE than 4,000 combinations
(tests/golden/gen_rule_agent.py caps the cost of the Python reference at 4,000).  They are the ones where the product's
branch and bound, path keys and teams do the work (k_auto2), so they are run through the REFERENCE's own
RuleBasedModel.choose (rule_based/utils/rule_based_model.py:43-101) here as well.  Same set-up as gen_rule_agent.py
(its header lists what is supplied for the import: the two spec-v1 stand-ins for the absent native decomposer functions,
an empty `tensorflow`, the numpy aliases, a 5-member env object)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_rule_agent as g8  # noqa: E402  (sets up the reference imports)
from oracle import oracle  # noqa: E402

N_CASES = 400


def timed_choice(args):
    t0 = time.time()
    a = g8.ref_choice(args)
    return a, time.time() - t0


def main():
    import multiprocessing as mp
    t0 = time.time()
    # lord leads with 20 cards / farmers early in the game are the heavy ones: many tables, few plies
    cases = g8.game_states(1500, 4, seed=41, auto_roles=0b111) + g8.game_states(600, 9, seed=42, auto_roles=0b101)
    stats = np.array([oracle.auto_choose(h, l if l.any() else None, f, r, want_stats=True)[1] for h, l, f, r in cases])
    heavy = np.flatnonzero(stats[:, 0] > 4000)
    print(f"{len(cases)} states, {len(heavy)} with more than 4000 combinations (max {stats[:, 0].max()}, "
          f"nodes max {stats[:, 1].max()})", flush=True)
    order = heavy[np.argsort(stats[heavy, 0])]
    # a spread over the heavy range: every k-th by size, the heaviest ones included
    pick = sorted(set(order[np.linspace(0, len(order) - 1, N_CASES).astype(int)].tolist()) | set(order[-25:].tolist()))
    sel = [cases[k] for k in pick]
    print("combinations of the picked states: min / median / max", int(stats[pick, 0].min()), int(np.median(stats[pick, 0])),
          int(stats[pick, 0].max()), flush=True)
    with mp.Pool(min(7, os.cpu_count() or 1)) as pool:
        res = []
        for k, r in enumerate(pool.imap(timed_choice, sel, chunksize=1)):
            res.append(r)
            if k % 50 == 0:
                print(f"  case {k}: {stats[pick[k], 0]} combinations, reference choose() -> {r[0]} in {r[1]:.1f} s", flush=True)
    choice = np.array([r[0] for r in res], np.int32)
    mine = np.array([oracle.auto_choose(h, l if l.any() else None, f, r) for h, l, f, r in sel], np.int32)
    bad = np.flatnonzero(mine != choice)
    print(f"oracle restatement vs reference choose(): {len(bad)} mismatches of {len(sel)} ({time.time() - t0:.0f}s)")
    for k in bad:
        print("  ", sel[k], "reference", choice[k], "oracle", mine[k])
    np.savez_compressed(os.path.join(HERE, "rule_agent_heavy.npz"),
                        hand=np.stack([c[0] for c in sel]), last=np.stack([c[1] for c in sel]),
                        left=np.stack([c[2] for c in sel]).astype(np.int8), role=np.array([c[3] for c in sel], np.int8),
                        choice=choice, combinations=stats[pick, 0], nodes=stats[pick, 1],
                        seconds=np.array([r[1] for r in res]))
    assert bad.size == 0
    print(f"wrote rule_agent_heavy.npz: {len(sel)} cases")


if __name__ == "__main__":
    main()
