"""CPU: pin the oracle (oracle/ddz_oracle.c) against the fixtures generated from the
reference's importable rules (tests/golden/gen_golden.py; card.py:34-159, :307-325,
:372-527; utils.py:45-63)."""
import numpy as np
import pytest

NA = 13527


def test_action_table_matches_card_py(oracle, golden):
    g = golden("action_table.npz")
    rows, info = oracle.action_table()
    assert rows.shape == (NA, 16) and g["rows"].shape == (NA, 15)
    assert np.array_equal(rows[:, :15], g["rows"])
    assert np.array_equal(rows[:, 15].astype(np.uint8), g["cat_range"])
    # CardGroup.to_cardgroup(row) agrees with the range the row was generated in
    assert np.array_equal(info[:, 0], g["tg_type"])
    assert np.array_equal(info[:, 1], g["tg_value"])
    assert np.array_equal(info[:, 2], g["tg_len"])
    assert np.array_equal(info[:, 3], g["rows"].sum(1))
    c2r = g["category2range"]
    assert c2r[-1, 1] == NA and list(np.diff(c2r, axis=1)[:, 0]) == [
        1, 15, 13, 13, 13, 182, 156, 36, 52, 45, 8033, 2939, 1, 1170, 858]


def test_lookup_roundtrip(oracle):
    rows, _ = oracle.action_table()
    for j in list(range(0, NA, 97)) + [NA - 1]:
        assert oracle.lookup(rows[j, :15]) == j
    bad = np.zeros(15, np.int8); bad[0] = 2; bad[1] = 1  # 3 3 4 is no combo
    assert oracle.lookup(bad) == -1


def test_beats_matches_bigger_than(oracle, golden):
    g = golden("beats.npz")
    bits = np.unpackbits(g["bits"], axis=1)[:, :NA]
    L = oracle.lib()
    for i, lid in enumerate(g["last_ids"]):
        mine = np.fromiter((L.ddzo_beats(j, int(lid)) for j in range(NA)), np.uint8, NA)
        assert np.array_equal(mine, bits[i]), f"last id {lid}"


def test_legal_cases_match_get_mask(oracle, golden):
    g = golden("legal_cases.npz")
    rows, _ = oracle.action_table()
    n = len(g["hands"])
    assert n > 900
    seen_cats = set()
    for k in range(n):
        lid = int(g["last_ids"][k])
        seen_cats.add(int(rows[lid, 15]))
        last = None if lid == 0 else rows[lid, :15]
        mine = oracle.legal(g["hands"][k], last)
        ref = g["ids"][g["offsets"][k]:g["offsets"][k + 1]].astype(np.int32)
        assert np.array_equal(mine, ref), f"case {k}"
    assert seen_cats == set(range(15))


def test_legal_sweep_matches_get_mask(oracle, golden):
    """G6: 4,000 uniformly random (hand, last) pairs, `last` uniform over the whole action space, legal ids by
    the reference's get_mask_onehot60."""
    g = golden("legal_sweep.npz")
    rows, _ = oracle.action_table()
    n = len(g["hands"])
    assert n == 4000 and (g["last_ids"] == 0).sum() == 800
    for k in range(n):
        lid = int(g["last_ids"][k])
        mine = oracle.legal(g["hands"][k], None if lid == 0 else rows[lid, :15])
        assert np.array_equal(mine, g["ids"][g["offsets"][k]:g["offsets"][k + 1]].astype(np.int32)), f"case {k}"


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, out in kat:
        assert list(oracle.philox(ctr, key)) == out


def test_thermometer_matches_char2onehot60(oracle, golden):
    bits = np.unpackbits(golden("thermo.npz")["bits"], axis=1)[:, :60]
    rows, _ = oracle.action_table()
    oh = oracle.rows_to_onehot(rows).reshape(NA, 60)
    assert np.array_equal(oh.astype(np.uint8), bits)


def test_episodes_replay(oracle, golden):
    """G4: the oracle env reproduces the committed seeded trajectories, whose per-ply
    legal sets were checked against the reference mask when the fixture was made."""
    g = golden("episodes.npz")
    T, iters = int(g["n_tables"]), int(g["n_iters"])
    env = oracle.OracleEnv(T, seed=int(g["seed"]), gid_base=int(g["gid_base"]))
    env.reset()
    k = 0
    for it in range(iters):
        offsets, rows, ids = env.legal()
        offsets = offsets.copy(); ids = ids.copy()
        meta = env.field(10).copy()
        done, reward, illegal, traj = env.step(oracle.STEP_RANDOM, auto_reset=True, want_traj=True)
        for t in range(T):
            seg = ids[offsets[t]:offsets[t + 1]]
            ref = g["ids"][g["offsets"][k]:g["offsets"][k + 1]]
            assert np.array_equal(seg, ref)
            assert meta[t, 0] == g["role"][k]
            assert int(traj[t, 28:32].view(np.int32)[0]) == g["choice"][k]
            assert done[t] == g["done"][k] and reward[t] == g["reward"][k]
            k += 1
    assert np.array_equal(env.state, g["final_state"])
    assert g["done"].sum() >= 6  # several full episodes, auto-reset exercised


def test_deal_is_a_partition_of_the_deck(oracle):
    env = oracle.OracleEnv(512, seed=7, gid_base=0)
    env.reset()
    hands = np.stack([env.field(r)[:, :15].astype(np.int32) for r in range(3)])
    assert np.array_equal(hands.sum(0), np.tile([4] * 13 + [1, 1], (512, 1)))
    assert np.array_equal(hands.sum(2).T, np.tile([17, 20, 17], (512, 1)))
    left = np.stack([env.field(r)[:, 15] for r in range(3)]).T
    assert np.array_equal(left, np.tile([17, 20, 17], (512, 1)))
    assert (env.field(10)[:, 0] == 1).all()  # lord moves first (game.py:173)
    # table ids key the RNG: shard [256, 512) of a second env deals the same cards
    env2 = oracle.OracleEnv(256, seed=7, gid_base=256)
    env2.reset()
    assert np.array_equal(env2.field(1), env.field(1)[256:])
    # episodes differ
    a = env.field(1).copy(); env.reset()
    assert not np.array_equal(a, env.field(1))


def test_observe_planes(oracle):
    env = oracle.OracleEnv(64, seed=3)
    env.reset()
    for _ in range(7):
        env.legal(); env.step(oracle.STEP_RANDOM)
    for variant, P in enumerate(oracle.PLANES):
        face = env.observe(variant)
        assert face.shape == (64, P, 15, 4)
        meta = env.field(10)
        for t in range(0, 64, 9):
            role = int(meta[t, 0])
            hand = env.field(role)[t, :15]
            assert np.array_equal(face[t, 0].sum(1), hand)           # onehot2arr (envi.py:148-157)
            assert np.array_equal(face[t, 1].sum(1), env.field(9)[t, :15])
            n1 = int(env.field((role + 1) % 3)[t, 15]); n2 = int(env.field((role + 2) % 3)[t, 15])
            unseen = np.array([4] * 13 + [1, 1]) - hand - env.field(9)[t, :15]
            assert np.allclose(face[t, P - 2].sum(1), unseen * np.float32(n1) / np.float32(n1 + n2))
            assert np.allclose(face[t, P - 2] + face[t, P - 1], (face[t, P - 2] > 0).astype(np.float32))


def test_step_rows_and_illegal(oracle):
    env = oracle.OracleEnv(8, seed=11)
    env.reset()
    offsets, rows, ids = env.legal()
    sel = np.zeros((8, 16), np.int8)
    for t in range(8):
        sel[t] = rows[offsets[t + 1] - 1]      # last legal row of each table
    sel[3, :15] = 0; sel[3, 0] = 3; sel[3, 1] = 1; sel[3, 2] = 1  # 333 4 5: never a combo
    state0 = env.state.copy()
    done, reward, illegal, _ = env.step(oracle.STEP_ROWS, sel, auto_reset=False)
    assert illegal.tolist() == [0, 0, 0, 1, 0, 0, 0, 0]
    f0 = state0.reshape(8, 11, 16); f1 = env.state.reshape(8, 11, 16)
    assert np.array_equal(f0[3], f1[3])  # illegal table untouched
    assert (f1[[0, 1, 2, 4], 10, 0] == 2).all()  # others advanced lord -> down
    # choice out of range is illegal too
    env.legal()
    done, reward, illegal, _ = env.step(oracle.STEP_CHOICE, np.full(8, 10 ** 6, np.int32), auto_reset=False)
    assert illegal.all()


def test_rollout_threads_equal_single_thread():
    """bench.py's all-core CPU baseline: splitting the tables over threads changes nothing
    (RNG keyed by the global table id)."""
    from oracle import oracle
    a = oracle.OracleEnv(96, seed=5)
    b = oracle.OracleEnv(96, seed=5)
    a.reset(); b.reset()
    ra = a.rollout_random(40)
    rb = oracle.rollout_random_mt(b, 40, 5)
    assert ra == rb
    assert np.array_equal(a.state, b.state)
