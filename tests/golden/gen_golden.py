#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's own
importable rules.  Runs ONLY in the build container (it imports
/root/reference/rule_based/utils/card.py and server/rule_utils/utils.py); the
fixtures it writes are plain data (inputs + expected outputs), never reference
source.  Usage:  python tests/golden/gen_golden.py [--quick]

  G1 action_table.npz  the 13,527 rows of card.py:get_action_space() with the
                       category range each falls in (card.py Category2Range)
                       and (type, value, len) of CardGroup.to_cardgroup(row)
  G2 beats.npz         CardGroup.bigger_than(row_j, last) for sampled `last`
                       rows x all rows, bit-packed
  G3 legal_cases.npz   (hand, last) -> legal ids from the reference's
                       get_mask_onehot60 (server/rule_utils/utils.py:20-38),
                       pass handled as get_mask does (rule_based/utils/utils.py:53-55)
  G4 episodes.npz      seeded random-policy episodes of the ORACLE env, every
                       ply's legal set cross-checked against the reference mask
                       at generation time (deal / RNG are spec v1 of this repo)
  G5 thermo.npz        Card.char2onehot60 (card.py:184-192) of every action row
  G6 legal_sweep.npz   4,000 uniformly random cases -- hands of 1..20 random cards, `last` uniform over the whole
                       action space (or a lead, 1 in 5) -- through the same reference functions as G3
"""
import argparse
import os
import sys
import time

import numpy as np

sys.dont_write_bytecode = True  # never write into /root/reference
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import rule_based.utils.card as rcard  # noqa: E402  (reference, container only)
from server.rule_utils.utils import get_mask_onehot60  # noqa: E402
import server.rule_utils.card as scard  # noqa: E402

CARDS = rcard.Card.cards
NA = len(rcard.action_space)


def to_counts(chars):
    c = np.zeros(15, np.int8)
    for ch in chars:
        c[CARDS.index(ch)] += 1
    return c


def to_chars(counts):
    out = []
    for i, n in enumerate(counts):
        out += [CARDS[i]] * int(n)
    return out


def ref_legal_ids(hand_counts, last_id):
    """legal ids by the reference's get_mask_onehot60 (+ get_mask's pass rule)."""
    hand = to_chars(hand_counts)
    last = None if last_id == 0 else scard.action_space[last_id]
    mask = get_mask_onehot60(hand, scard.action_space, last)
    legal = np.flatnonzero(mask[1:].sum(axis=1) > 0) + 1
    if last_id != 0 and len(hand) > 0:
        legal = np.concatenate([[0], legal])  # mask[0] stays 1 when following
    return legal.astype(np.int32)


def gen_g1():
    rows = np.stack([to_counts(a) for a in rcard.action_space])
    cat_range = np.zeros(NA, np.uint8)
    for cat, (lo, hi) in enumerate(rcard.Category2Range):
        cat_range[lo:hi] = cat
    tg = np.zeros((NA, 3), np.int32)
    groups = []
    for j, a in enumerate(rcard.action_space):
        g = rcard.CardGroup.to_cardgroup(a)
        groups.append(g)
        tg[j] = (g.type, g.value, g.len)
    np.savez_compressed(os.path.join(HERE, "action_table.npz"), rows=rows, cat_range=cat_range,
                        tg_type=tg[:, 0].astype(np.uint8), tg_value=tg[:, 1].astype(np.uint8),
                        tg_len=tg[:, 2].astype(np.uint8),
                        category2range=np.array(rcard.Category2Range, np.int32))
    return rows, cat_range, groups


def gen_g2(groups, cat_range, rng, per_cat):
    last_ids = []
    for cat in range(15):
        ids = np.flatnonzero(cat_range == cat)
        k = min(len(ids), per_cat)
        last_ids += sorted(rng.choice(ids, k, replace=False).tolist())
    last_ids = np.array(last_ids, np.int32)
    bits = np.zeros((len(last_ids), NA), np.uint8)
    for i, lid in enumerate(last_ids):
        g = groups[lid]
        for j in range(NA):
            bits[i, j] = groups[j].bigger_than(g)
    np.savez_compressed(os.path.join(HERE, "beats.npz"), last_ids=last_ids,
                        bits=np.packbits(bits, axis=1))
    return last_ids


def random_hand(rng, n):
    deck = np.array([r for r in range(13) for _ in range(4)] + [13, 14])
    pick = rng.choice(deck, n, replace=False)
    return np.bincount(pick, minlength=15).astype(np.int8)


ADVERSARIAL = [
    "3334445556667778910JQ", "33334444555566667777", "3333444455556666*$", "345678910JQKA",
    "3344556677889910JJQQ", "333444555666777888", "KKKAAA22*$", "2222AAAAKKKK*$",
    "33344455566677789JQK", "3456789910JQKA2*$", "3", "*$", "2222", "*", "$", "33", "333",
    "3334", "33344", "34567", "334455", "333444", "33344456", "3334445566",
    "33334445", "3333444555", "3333*$", "333444*$", "333444555*$6", "44445555666677",
    "JJJQQQKKKAAA2222*$", "10101010JJJJQQQQKKKK", "33445566778899", "334455667788991010JJ",
    "333444555666777888", "3334445556667778", "555666777888999", "33344455566678910J",
]


def parse_hand(s):
    out, i = [], 0
    while i < len(s):
        if s[i] == "1":
            out.append("10"); i += 2
        else:
            out.append(s[i]); i += 1
    return to_counts(out)


def gen_g3(rows, cat_range, rng, n_random):
    hands, lasts = [], []
    for s in ADVERSARIAL:  # lead on every adversarial hand
        hands.append(parse_hand(s)); lasts.append(0)
    for n in range(1, 21):  # leads at every hand size
        for _ in range(max(1, n_random // 60)):
            hands.append(random_hand(rng, n)); lasts.append(0)
    # follows: every category as `last`, several rows each, random + adversarial hands
    for cat in range(1, 15):
        ids = np.flatnonzero(cat_range == cat)
        for lid in rng.choice(ids, min(len(ids), max(4, n_random // 40)), replace=False):
            n = int(rng.integers(max(1, int(rows[lid].sum()) - 2), 21))
            hands.append(random_hand(rng, n)); lasts.append(int(lid))
        for s in ADVERSARIAL[:12]:
            hands.append(parse_hand(s)); lasts.append(int(rng.choice(ids)))
    # follows where the hand is the full deck minus `last` (everything that can beat it)
    full = np.array([4] * 13 + [1, 1], np.int8)
    for cat in range(1, 15):
        ids = np.flatnonzero(cat_range == cat)
        for lid in rng.choice(ids, min(len(ids), 3), replace=False):
            hands.append(full - rows[lid]); lasts.append(int(lid))
    hands.append(full); lasts.append(0)
    hands.append(np.zeros(15, np.int8)); lasts.append(0)      # empty hand, lead
    hands.append(np.zeros(15, np.int8)); lasts.append(1)      # empty hand, follow
    offs, ids = [0], []
    t0 = time.time()
    for k, (h, l) in enumerate(zip(hands, lasts)):
        leg = ref_legal_ids(h, l)
        ids.append(leg)
        offs.append(offs[-1] + len(leg))
        if k % 100 == 0:
            print(f"  G3 {k}/{len(hands)}  {time.time() - t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "legal_cases.npz"), hands=np.stack(hands),
                        last_ids=np.array(lasts, np.int32), offsets=np.array(offs, np.int32),
                        ids=np.concatenate(ids).astype(np.int16))
    return len(hands)


def gen_g4(n_tables, n_iters, seed, gid_base):
    from oracle import oracle
    env = oracle.OracleEnv(n_tables, seed=seed, gid_base=gid_base)
    env.reset()
    rec = {k: [] for k in ("iter", "table", "episode", "ply", "role", "hand", "last_id", "nlegal",
                           "choice", "action_id", "done", "reward")}
    offs, ids = [0], []
    for it in range(n_iters):
        offsets, rows, lids = env.legal()
        offsets = offsets.copy(); rows = rows.copy(); lids = lids.copy()
        meta = env.field(10).copy()
        hands = [env.field(r).copy() for r in range(3)]
        recent = [env.field(6 + r).copy() for r in range(3)]
        done, reward, illegal, traj = env.step(oracle.STEP_RANDOM, auto_reset=True, want_traj=True)
        assert not illegal.any()
        for t in range(n_tables):
            role = int(meta[t, 0])
            hand = hands[role][t, :15].astype(np.int8)
            b1 = recent[(role + 2) % 3][t, :15].astype(np.int8)
            b2 = recent[(role + 1) % 3][t, :15].astype(np.int8)
            last = b1 if b1.any() else b2
            last_id = oracle.lookup(last)
            seg = lids[offsets[t]:offsets[t + 1]]
            ref = ref_legal_ids(hand, last_id)
            assert np.array_equal(seg, ref), (it, t, hand, last_id, seg, ref)
            choice = int(np.frombuffer(traj[t, 28:32].tobytes(), np.int32)[0])
            rec["iter"].append(it); rec["table"].append(t)
            rec["episode"].append(int(np.frombuffer(meta[t, 8:12].tobytes(), np.uint32)[0]))
            rec["ply"].append(int(np.frombuffer(meta[t, 4:6].tobytes(), np.uint16)[0]))
            rec["role"].append(role); rec["hand"].append(hand); rec["last_id"].append(last_id)
            rec["nlegal"].append(len(seg)); rec["choice"].append(choice)
            rec["action_id"].append(int(seg[choice]))
            rec["done"].append(int(done[t])); rec["reward"].append(int(reward[t]))
            ids.append(seg); offs.append(offs[-1] + len(seg))
        print(f"  G4 iter {it + 1}/{n_iters}", flush=True)
    out = {k: np.array(v) for k, v in rec.items()}
    out["hand"] = np.stack(rec["hand"]).astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "episodes.npz"), n_tables=n_tables, n_iters=n_iters,
                        seed=np.uint64(seed), gid_base=np.uint64(gid_base),
                        offsets=np.array(offs, np.int32),
                        ids=np.concatenate(ids).astype(np.int16),
                        final_state=env.state.copy(), **out)


def gen_g5(rows):
    bits = np.stack([rcard.Card.char2onehot60(a) for a in rcard.action_space]).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "thermo.npz"), bits=np.packbits(bits, axis=1))


def _g6_case(args):
    return ref_legal_ids(*args)


def gen_g6(rows, n_cases, seed):
    import multiprocessing as mp
    rng = np.random.default_rng(seed)
    hands = np.stack([random_hand(rng, int(rng.integers(1, 21))) for _ in range(n_cases)])
    lasts = rng.integers(1, NA, n_cases).astype(np.int32)
    lasts[np.arange(n_cases) % 5 == 0] = 0
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        legal = pool.map(_g6_case, [(hands[i], int(lasts[i])) for i in range(n_cases)], chunksize=25)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in legal])]).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "legal_sweep.npz"), hands=hands, last_ids=lasts, offsets=offs,
                        ids=np.concatenate(legal).astype(np.int16))
    return n_cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="small fixture set (smoke of this script)")
    ap.add_argument("--only", default="", help="comma list of g1..g5")
    a = ap.parse_args()
    only = set(a.only.split(",")) if a.only else {"g1", "g2", "g3", "g4", "g5", "g6"}
    rng = np.random.default_rng(20261004)
    t0 = time.time()
    rows, cat_range, groups = gen_g1()
    print(f"G1 action table: {NA} rows ({time.time() - t0:.0f}s)", flush=True)
    if "g2" in only:
        lids = gen_g2(groups, cat_range, rng, 3 if a.quick else 40)
        print(f"G2 beats: {len(lids)} last rows ({time.time() - t0:.0f}s)", flush=True)
    if "g3" in only:
        n = gen_g3(rows, cat_range, rng, 60 if a.quick else 1200)
        print(f"G3 legal cases: {n} ({time.time() - t0:.0f}s)", flush=True)
    if "g4" in only:
        gen_g4(2 if a.quick else 6, 12 if a.quick else 150, seed=0x5EED0001, gid_base=1000)
        print(f"G4 episodes ({time.time() - t0:.0f}s)", flush=True)
    if "g5" in only:
        gen_g5(rows)
        print(f"G5 thermometer ({time.time() - t0:.0f}s)", flush=True)
    if "g6" in only:
        n = gen_g6(rows, 200 if a.quick else 4000, seed=20261005)
        print(f"G6 legal sweep: {n} cases ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
