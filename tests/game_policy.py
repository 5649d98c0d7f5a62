"""Deterministic stand-in policies shared by tests/golden/gen_game.py (which drives the REFERENCE's Game.play with them)
and the tests that replay fixture G11 through the batched engine: a choice is a function of the observation's integer
planes and the list size only, so the batched replay reproduces it without any RNG.

face plane 0 = the actor's hand, plane 1 = the cards taken so far (envi.py:87-96: the first two planes of every Env
class).  e_greedy stands for DQNFirst.e_greedy_action (dqn.py:50-61), greedy for DQNFirst.greedy_action (:63-71)."""


def e_greedy_index(hand_cards, taken_cards, n_actions):
    return (5 * hand_cards + 3 * taken_cards + 1) % n_actions


def greedy_index(hand_cards, taken_cards, n_actions):
    return (hand_cards + 2 * taken_cards) % n_actions


# the three set-ups of Game() the fixture holds (game.py:11-43): which roles have a network (the others are played by
# Env.step_auto, game.py:106), which of those keep training (game.py:95-104: e_greedy + s0/a0 bookkeeping, else greedy
# and no feedback), the Env class's face variant, and the seed of the tables
SCENARIOS = {
    "all_trained":    {"ai": ("lord", "down", "up"), "train": ("lord", "down", "up"), "variant": 3, "seed": 2026},
    "lord_vs_rule":   {"ai": ("lord",), "train": ("lord",), "variant": 0, "seed": 2027},
    "farmers_train":  {"ai": ("lord", "down", "up"), "train": ("down", "up"), "variant": 2, "seed": 2028},
}
ROLE_ID = {"up": 0, "lord": 1, "down": 2}     # envi.py:24,47-58


def _thermo(rows15):
    import numpy as np
    return (np.asarray(rows15)[..., None] > np.arange(4)).astype(np.float32)


def replay_scenario(backend, glue, name, tables, episodes, quirk):
    """Drive `backend` (T tables in lock step: the oracle env on the CPU, BatchedEnv on the GPU) through the scenario
    with the deterministic policies above and dqn_glue.TransitionAssembler; returns per table the list of closed
    transitions (role, reward, done, crc32(s0), crc32(s1), a0 counts, a1 counts) in emission order, and the wins
    [T,3] (up, lord, down).  backend: object with roles() -> int[T], legal() -> (offsets, rows, ids) numpy CSR,
    observe(variant) -> f32 [T,P,15,4] numpy, auto_choose() -> int32[T] numpy, step_ids(ids) -> (done, r) numpy
    (no auto-reset), reset(mask)."""
    import zlib
    import numpy as np
    import torch
    sc = SCENARIOS[name]
    T = tables
    ai = np.zeros(3, bool)
    tr = np.zeros(3, bool)
    for r in sc["ai"]:
        ai[ROLE_ID[r]] = True
    for r in sc["train"]:
        tr[ROLE_ID[r]] = True
    variant = sc["variant"]
    P = (4, 7, 9, 6)[variant]
    asm = glue.TransitionAssembler(T, P, "cpu", trained_roles=tuple(tr), replicate_reference_quirk=quirk)
    got = [[] for _ in range(T)]
    wins = np.zeros((T, 3), np.int64)
    played = np.zeros(T, np.int64)

    def collect(d):
        for k in range(d["table"].numel()):
            t = int(d["table"][k])
            got[t].append((int(d["role"][k]), float(d["reward"][k]), bool(d["done"][k]),
                           zlib.crc32(d["s0"][k].numpy().tobytes()), zlib.crc32(d["s1"][k].numpy().tobytes()),
                           d["a0"][k].numpy().sum(1).astype(np.int8), d["a1"][k].numpy().sum(1).astype(np.int8)))

    backend.reset(None)
    guard = 0
    while (played < episodes).any():
        guard += 1
        assert guard < 200 * episodes
        active = played < episodes
        role = np.asarray(backend.roles()).astype(np.int64)
        offsets, rows, ids = backend.legal()
        face = backend.observe(variant)
        n = np.diff(offsets).astype(np.int64)
        h = np.rint(face[:, 0].sum((1, 2))).astype(np.int64)
        tk = np.rint(face[:, 1].sum((1, 2))).astype(np.int64)
        nn = np.maximum(n, 1)
        gi = offsets[:-1] + greedy_index(h, tk, nn)
        ei = offsets[:-1] + e_greedy_index(h, tk, nn)
        pick = np.where(tr[role], ei, gi)                     # game.py:95-101: e_greedy while training, else greedy
        pick = np.where(active, pick, 0)
        gi = np.where(active, gi, 0)
        sel = np.where(ai[role], ids[np.minimum(pick, len(ids) - 1)], backend.auto_choose()).astype(np.int32)
        chosen, greedy = _thermo(rows[np.minimum(pick, len(ids) - 1), :15]), _thermo(rows[np.minimum(gi, len(ids) - 1), :15])
        collect(asm.before_step(torch.from_numpy(role), torch.from_numpy(face), torch.from_numpy(chosen),
                                torch.from_numpy(greedy), active=torch.from_numpy(active & ai[role])))
        done, r = backend.step_ids(sel)
        done = np.asarray(done).astype(bool) & active
        collect(asm.after_step(torch.from_numpy(role), torch.from_numpy(done.astype(np.uint8)),
                               torch.from_numpy(np.asarray(r).astype(np.int8)), torch.from_numpy(backend.observe(variant))))
        np.add.at(wins, (np.flatnonzero(done), role[done]), 1)
        played += done
        backend.reset((done & (played < episodes)).astype(np.uint8))
    return got, wins


def expected_from_fixture(g, name, tables, quirk, trained_roles):
    """fixture G11 -> per table the reference's perceive() calls; quirk=False drops the stale transitions: the first
    non-terminal feedback of `down` and of `up` in every episode but the first of a Game object (game.py:132,147 find
    *_a0 of the previous episode)."""
    want = [[] for _ in range(tables)]
    tb, ro, ep = g[f"{name}.table"], g[f"{name}.role"], g[f"{name}.episode"]
    seen = set()
    for k in range(len(tb)):
        key = (int(tb[k]), int(ep[k]), int(ro[k]))
        first = key not in seen
        seen.add(key)
        stale = first and ep[k] > 0 and ro[k] in (0, 2)
        if stale:
            assert not g[f"{name}.done"][k] and g[f"{name}.reward"][k] == 0
            if not quirk:
                continue
        want[int(tb[k])].append((int(ro[k]), float(g[f"{name}.reward"][k]), bool(g[f"{name}.done"][k]),
                                 int(g[f"{name}.s0_crc"][k]), int(g[f"{name}.s1_crc"][k]), g[f"{name}.a0"][k], g[f"{name}.a1"][k]))
    return want
