"""CPU: TransitionAssembler (batched game.py:109-167 feedback bookkeeping) against a plain
per-table restatement of the reference's play loop, on the oracle env with a deterministic
policy (play the last legal move; greedy = the first legal move)."""
import importlib

import numpy as np
import torch

P_VARIANT, PLANES = 3, 6
REWARD = {0: 50.0, 1: 100.0, 2: 50.0}


def _onehot(row15):
    return (np.asarray(row15)[:, None] > np.arange(4)[None, :]).astype(np.float32)


def _reference_loop(oracle, seed, n_plies):
    """one table, the control flow of Game.play / step / feedback (game.py:90-181), episodes
    played back to back; returns the list of closed transitions in order."""
    env = oracle.OracleEnv(1, seed=seed)
    env.reset()
    pend = {}
    out = []
    for _ in range(n_plies):
        offsets, rows, _ = env.legal()
        role = int(env.field(10)[0, 0])
        face = env.observe(P_VARIANT)[0].copy()
        chosen, greedy = _onehot(rows[offsets[1] - 1, :15]), _onehot(rows[0, :15])
        if role in pend:                                   # feedback(role, done=False): game.py:109-127
            s0, a0 = pend[role]
            out.append((0, role, s0, a0, 0.0, face, greedy, False))
        pend[role] = (face, chosen)                        # step(): game.py:95-104
        done, r, _, _ = env.step(oracle.STEP_CHOICE, np.array([offsets[1] - 1], np.int32), auto_reset=False)
        if done[0]:                                        # terminal feedback: game.py:134-141, :149-167
            tface = env.observe(P_VARIANT)[0].copy()
            lord_won = r[0] < 0
            for x in sorted(pend, key=lambda x: (x - role - 1) % 3):   # the reference's call order (game.py:134-167)
                s0, a0 = pend[x]
                rew = REWARD[x] if (x == 1) == lord_won else -REWARD[x]
                out.append((0, x, s0, a0, rew, tface, np.zeros((15, 4), np.float32), True))
            pend = {}
            env.reset()
    return out


def test_assembler_matches_per_table_loop(oracle):
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    T, n = 5, 230
    seeds = [11, 12, 13, 14, 15]
    want = {t: _reference_loop(oracle, seeds[t], n) for t in range(T)}
    envs = [oracle.OracleEnv(1, seed=s) for s in seeds]   # T independent tables driven batch-wise
    for e in envs:
        e.reset()
    asm = glue.TransitionAssembler(T, PLANES, "cpu")
    got = {t: [] for t in range(T)}

    def collect(tr):
        for k in range(tr["table"].numel()):
            t = int(tr["table"][k])
            got[t].append((0, int(tr["role"][k]), tr["s0"][k].numpy(), tr["a0"][k].numpy(), float(tr["reward"][k]),
                           tr["s1"][k].numpy(), tr["a1"][k].numpy(), bool(tr["done"][k])))

    for _ in range(n):
        roles, faces, chosen, greedy, last = [], [], [], [], []
        for e in envs:
            offsets, rows, _ = e.legal()
            roles.append(int(e.field(10)[0, 0])); faces.append(e.observe(P_VARIANT)[0])
            chosen.append(_onehot(rows[offsets[1] - 1, :15])); greedy.append(_onehot(rows[0, :15]))
            last.append(offsets[1] - 1)
        role = torch.tensor(roles)
        collect(asm.before_step(role, torch.from_numpy(np.stack(faces)), torch.from_numpy(np.stack(chosen)),
                                torch.from_numpy(np.stack(greedy))))
        dones, rs, tfaces = [], [], []
        for e, idx in zip(envs, last):
            d, r, _, _ = e.step(oracle.STEP_CHOICE, np.array([idx], np.int32), auto_reset=False)
            dones.append(int(d[0])); rs.append(int(r[0])); tfaces.append(e.observe(P_VARIANT)[0])
        done = torch.tensor(dones, dtype=torch.uint8)
        collect(asm.after_step(role, done, torch.tensor(rs, dtype=torch.int8), torch.from_numpy(np.stack(tfaces))))
        for e, d in zip(envs, dones):
            if d:
                e.reset()
    n_done = 0
    for t in range(T):
        assert len(got[t]) == len(want[t]) > 150
        for a, b in zip(got[t], want[t]):
            assert a[1] == b[1] and a[4] == b[4] and a[7] == b[7]
            for i in (2, 3, 5, 6):
                assert np.array_equal(a[i], b[i])
            n_done += a[7]
    assert n_done >= 9                                  # several finished episodes, 3 transitions each
    tr = {"reward": torch.tensor([0.0, 100.0]), "done": torch.tensor([False, True])}
    assert torch.allclose(glue.td_target(tr, torch.tensor([2.0, 7.0])), torch.tensor([1.9, 100.0]))


# ---- the ragged Q forward, factorised (dqn_glue.QNet / FactorisedQ / ragged_q) ----
REF_NET_SHAPES = {  # net.py:137-150 NetCooperationSimplify (7 input planes); the other Net* classes differ in C only
    "conv1.weight": (256, 7, 1, 1), "conv2.weight": (256, 7, 1, 2), "conv3.weight": (256, 7, 1, 3),
    "conv4.weight": (256, 7, 1, 4), "conv_shunzi.weight": (256, 7, 15, 1), "fc1.weight": (256, 4864),
    "fc2.weight": (1, 256), "conv1.bias": (256,), "conv2.bias": (256,), "conv3.bias": (256,), "conv4.bias": (256,),
    "conv_shunzi.bias": (256,), "fc1.bias": (256,), "fc2.bias": (1,)}


def _random_faces_and_rows(T, P, seed):
    g = torch.Generator().manual_seed(seed)
    face = (torch.rand(T, P, 15, 4, generator=g) < 0.4).float()
    face[:, -2:] *= torch.rand(T, 1, 1, 1, generator=g)               # the two probability planes hold fractions
    counts = torch.randint(0, 12, (T,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]).int()
    N = int(offsets[-1])
    rows = torch.zeros((N + 7, 16), dtype=torch.int8)                 # 7 padding rows behind offsets[T]
    rows[:, :15] = ((torch.rand(N + 7, 15, generator=g) < 0.2) * torch.randint(1, 5, (N + 7, 15), generator=g)).to(torch.int8)
    rows[:, 13:15].clamp_(max=1)                                      # a joker exists once (count rows of the action space)
    rows[0, :15] = 0                                                  # a pass
    return face, rows, offsets, counts, N


def test_qnet_is_state_dict_compatible_with_the_reference():
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    sd = glue.QNet(6).state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == REF_NET_SHAPES
    for planes, c in ((4, 5), (7, 8), (9, 10)):                       # net.py:66-134: the other three variants
        assert glue.QNet(planes).conv4.weight.shape == (256, c, 1, 4)


def test_factorised_q_equals_literal_conv_evaluation():
    """ragged_q (first layer factorised per (rank, count), conv_shunzi folded through fc1) == the literal evaluation of
    net.py:81-102 on face repeated per action + the action plane.  Floating point, fp32: tolerance 1e-5 (absolute, on
    outputs of magnitude ~0.1: the two differ in summation order only)."""
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    for P in (4, 7, 9, 6):
        torch.manual_seed(P)
        net = glue.QNet(P).eval()
        face, rows, offsets, counts, N = _random_faces_and_rows(41, P, 100 + P)
        q = glue.ragged_q(net, face, rows, offsets)
        assert q.shape == (N + 7,) and bool(torch.isfinite(q).all())
        seg = torch.repeat_interleave(torch.arange(41), counts)
        acts = (rows[:N, :15].float()[:, :, None] > torch.arange(4)[None, None, :]).float()   # envi.py:139-146
        with torch.no_grad():
            want = net(face[seg], acts)[:, 0]
        assert float((q[:N] - want).abs().max()) < 1e-5
        # single state, the reference's own calling convention: face [P,15,4] repeated inside forward (net.py:87-88)
        with torch.no_grad():
            one = net(face[3], acts[seg == 3])[:, 0]
        assert float((q[:N][seg == 3] - one).abs().max()) < 1e-5


def test_packed_rows_hold_what_a_legal_move_can_use_and_give_the_same_q():
    """FactorisedQ.pack / tables_packed / q_csr_packed (the layout of ddz_q_features_packed / ddz_q_slab_packed): of a
    table's 69 (rank, count) rows only count 0 of every rank and counts 1..hands[t][r] exist; every rank's rows are one
    contiguous segment that starts with the T count-0 rows; q over the packed rows == q over the full tables (fp32,
    1e-5: the GEMM's row count is the only difference) for rows that respect the hands -- and == the literal network."""
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    for P in (6, 4):
        torch.manual_seed(20 + P)
        net = glue.QNet(P).eval()
        T = 37
        face, rows, offsets, counts, N = _random_faces_and_rows(T, P, 300 + P)
        g = torch.Generator().manual_seed(P)
        hands = torch.randint(0, 5, (T, 15), generator=g)
        hands[:, 13:].clamp_(max=1)
        hands[0] = 0                                                   # a table that holds nothing: 15 rows
        seg = torch.repeat_interleave(torch.arange(T), counts)
        rows[:N, :15] = torch.minimum(rows[:N, :15].long(), hands[seg]).to(torch.int8)   # moves take what the hand holds
        for batched in (False, True):   # fifteen exact segments / equally long padded ones (one batched GEMM)
            fq = glue.FactorisedQ(net)
            fq.batched_gemm = batched
            row_index, row0 = fq.pack(hands)
            assert row_index.shape == (T, 64) and row_index.dtype == torch.int32 and len(row0) == 16
            held = torch.zeros((T, 64), dtype=torch.bool)
            for r in range(15):
                for c in range(1, 5 if r < 13 else 2):
                    held[:, 4 * r + c - 1 if r < 13 else 52 + r - 13] = hands[:, r] >= c
            assert bool(((row_index >= 0) == held).all())
            held_r = hands.clamp(max=4).sum(0)                             # held (count >= 1) rows per rank
            if batched:
                M = row0[1]
                assert M % 2048 == 0 and row0 == [r * M for r in range(16)] and T + int(held_r.max()) <= M < T + int(held_r.max()) + 2048
            else:
                assert [row0[r + 1] - row0[r] for r in range(15)] == [T + int(x) for x in held_r] and row0[0] == 0
            used = row_index[row_index >= 0].long()
            assert used.unique().numel() == used.numel()                   # no row twice
            for r in range(15):                                            # rank r's held rows lie behind its T count-0 rows
                cols = slice(4 * r, 4 * r + 4) if r < 13 else slice(52 + r - 13, 53 + r - 13)
                v = row_index[:, cols][row_index[:, cols] >= 0]
                assert v.numel() == int(held_r[r]) and bool(((v >= row0[r] + T) & (v < row0[r] + T + int(held_r[r]))).all())
            pu = fq.tables_packed(face, hands, fused=False)
            qp = fq.q_csr_packed(pu, rows, offsets)
            qf = fq.q_csr(fq.tables(face, fused=False), rows, offsets)
            assert float((qp[:N] - qf[:N]).abs().max()) < 1e-5
            acts = (rows[:N, :15].float()[:, :, None] > torch.arange(4)[None, None, :]).float()
            with torch.no_grad():
                want = net(face[seg], acts)[:, 0]
            assert float((qp[:N] - want).abs().max()) < 1e-5


def test_needed_rows_form_equals_the_literal_network():
    """FactorisedQ.needed_torch / q_csr_needed (the statement of the engine's needed-rows kernels, csrc/ddz_qnet.h): H0 per
    table from one dense product + a D row only for the (rank, count >= 1) pairs some move of the table's list uses.  The
    layout: fifteen rank segments starting at multiples of the fc1 kernel's tile, inside a segment table-major then count, every needed
    (t, r, c) exactly one row, nothing else; q == the full factorised tables == the literal nn.Conv2d network (fp32, 1e-5
    absolute on outputs of magnitude ~0.1: summation order only)."""
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    for P in (6, 4, 7, 9):
        torch.manual_seed(40 + P)
        net = glue.QNet(P).eval()
        T = 53
        face, rows, offsets, counts, N = _random_faces_and_rows(T, P, 500 + P)
        fq = glue.FactorisedQ(net)
        nu = fq.needed_torch(face, rows, offsets)
        seg_t = torch.repeat_interleave(torch.arange(T), counts)
        # the need sets, brute force
        want_need = torch.zeros((T, 64), dtype=torch.bool)
        for j in range(N):
            for r in range(15):
                c = min(int(rows[j, r]), 4 if r < 13 else 1)
                if c > 0:
                    want_need[seg_t[j], 4 * r + c - 1 if r < 13 else 52 + r - 13] = True
        assert bool(((nu.row_index >= 0) == want_need).all())
        seg = nu.seg.tolist()
        used = nu.row_index[nu.row_index >= 0].long()
        assert used.unique().numel() == used.numel() == seg[32] == int(want_need.sum())
        TILE = glue.fc_tile()
        assert all(seg[r] % TILE == 0 and seg[16 + r] == seg[r] // TILE for r in range(16)) and seg[0] == 0
        for r in range(15):
            cols = slice(4 * r, 4 * r + 4) if r < 13 else slice(52 + r - 13, 53 + r - 13)
            v = nu.row_index[:, cols]
            n_r = int((v >= 0).sum())
            assert seg[r + 1] - seg[r] == (n_r + TILE - 1) // TILE * TILE
            got = v[v >= 0]
            assert got.tolist() == list(range(seg[r], seg[r] + n_r))       # table-major, then ascending count: consecutive
        q = fq.q_csr_needed(nu, rows, offsets)
        qf = fq.q_csr(fq.tables(face, fused=False), rows, offsets)
        assert float((q[:N] - qf[:N]).abs().max()) < 1e-5
        acts = (rows[:N, :15].float()[:, :, None] > torch.arange(4)[None, None, :]).float()
        with torch.no_grad():
            want = net(face[seg_t], acts)[:, 0]
        assert float((q[:N] - want).abs().max()) < 1e-5


def test_factorised_tables_follow_weight_updates_and_chunking():
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    torch.manual_seed(1)
    net = glue.QNet(6).eval()
    face, rows, offsets, counts, N = _random_faces_and_rows(23, 6, 5)
    q0 = glue.ragged_q(net, face, rows, offsets).clone()
    with torch.no_grad():
        net.fc2.bias += 0.5                                           # in-place update, as an optimizer step does
        net.conv3.weight.mul_(1.1)
    q1 = glue.ragged_q(net, face, rows, offsets)
    seg = torch.repeat_interleave(torch.arange(23), counts)
    acts = (rows[:N, :15].float()[:, :, None] > torch.arange(4)[None, None, :]).float()
    with torch.no_grad():
        want = net(face[seg], acts)[:, 0]
    assert float((q1[:N] - want).abs().max()) < 1e-5 and float((q1[:N] - q0[:N]).abs().max()) > 0.1
    small = glue.FactorisedQ(net, chunk_tables=5)                     # 23 tables in chunks of 5 (ragged last chunk)
    assert torch.allclose(small.tables(face), net._ddz_factorised.tables(face), rtol=0, atol=1e-6)


def test_replay_and_td_step():
    """dqn.py:21-48 on tensors: ring buffer, y = r + (1 - done) gamma Q_target(s1, a1), one Adam step moves the loss."""
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    torch.manual_seed(0)
    policy, target = glue.QNet(6), glue.QNet(6)
    target.load_state_dict(policy.state_dict())
    target.eval()
    opt = torch.optim.Adam(policy.parameters(), lr=1e-3)
    rp = glue.Replay(64, 6, "cpu")
    g = torch.Generator().manual_seed(3)
    for _ in range(5):
        k = 20
        rp.push({"s0": torch.rand(k, 6, 15, 4, generator=g), "a0": torch.rand(k, 15, 4, generator=g),
                 "s1": torch.rand(k, 6, 15, 4, generator=g), "a1": torch.rand(k, 15, 4, generator=g),
                 "reward": torch.randn(k, generator=g), "done": torch.rand(k, generator=g) < 0.3})
    assert rp.n == 64 and rp.head == 100 % 64
    b = rp.sample(32)
    y = glue.td_target(b, torch.ones(32), 0.95)
    assert torch.allclose(y, b["reward"] + (~b["done"]).float() * 0.95)
    policy.eval()                                                     # no dropout noise: the loss must go down
    l0 = float(glue.td_step(policy, target, opt, b))
    for _ in range(10):
        l1 = float(glue.td_step(policy, target, opt, b))
    assert l1 < l0


def test_qnet_and_ragged_q_against_the_references_own_networks(golden):
    """Fixture G9 (tests/golden/gen_qnet.py): q values computed by the reference's OWN classes (net.py:66-150, forward
    net.py:81-102) right after torch.manual_seed(seed), eval mode.  dqn_glue.QNet built after the same seed has the same
    parameters (same constructors in the same order: checked through per-parameter checksums, so a mismatch means
    "different initial weights", not "different function"), its literal forward reproduces the reference's q (same ops:
    1e-6), and the factorised inference form agrees to 1e-5 -- for all four networks, in both calling conventions
    (batch of (state, action) rows; one state with all its actions, dqn.py:56).  Plus the TD target of dqn.py:40-41."""
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    g = golden("qnet.npz")
    for P in (4, 7, 9, 6):
        torch.manual_seed(int(g[f"p{P}_seed"]))
        net = glue.QNet(P).eval()
        sd = net.state_dict()
        names = [str(x) for x in g[f"p{P}_param_names"]]
        assert names == sorted(sd) and [str(tuple(sd[k].shape)) for k in names] == [str(x) for x in g[f"p{P}_param_shapes"]]
        sums = np.array([[float(sd[k].double().sum()), float((sd[k].double() ** 2).sum())] for k in names])
        assert np.allclose(sums, g[f"p{P}_param_sums"], rtol=1e-9, atol=1e-9), "the seeded initial weights differ from the reference's"
        face, actions = torch.from_numpy(g[f"p{P}_face"]), torch.from_numpy(g[f"p{P}_actions"])
        want, want1 = torch.from_numpy(g[f"p{P}_q"]), torch.from_numpy(g[f"p{P}_q_single"])
        with torch.no_grad():
            q = net(face, actions)[:, 0]
            q1 = net(face[0], actions[: want1.numel()])[:, 0]
        assert float((q - want).abs().max()) < 1e-6 and float((q1 - want1).abs().max()) < 1e-6
        # the inference form: every (state, action) pair as a one-row segment; then one state with 7 actions
        n = face.shape[0]
        rows = torch.zeros((n, 16), dtype=torch.int8)
        rows[:, :15] = actions.sum(dim=2).round().to(torch.int8)          # onehot2arr (envi.py:148-157)
        offsets = torch.arange(n + 1, dtype=torch.int32)
        assert float((glue.ragged_q(net, face, rows, offsets) - want).abs().max()) < 1e-5
        m = want1.numel()
        off1 = torch.tensor([0, m], dtype=torch.int32)
        assert float((glue.ragged_q(net, face[:1], rows[:m], off1) - want1).abs().max()) < 1e-5
    batch = {"reward": torch.from_numpy(g["td_r"]), "done": torch.from_numpy(g["td_done"]) > 0}
    y = glue.td_target(batch, torch.from_numpy(g["td_qnext"]), float(g["gamma"]))
    assert torch.allclose(y, torch.from_numpy(g["td_y"]), rtol=0, atol=1e-6)


def test_td_step_against_the_references_own_perceive(golden):
    """Fixture G10 (tests/golden/gen_qnet.py): ONE DQNFirst.perceive() update of the reference's own agent (dqn.py:21-48:
    256 transitions sampled with Python's random, target = r + (1 - done) * GAMMA * Q_target(s1, a1), MSE, Adam 1e-4; both
    networks in train mode, i.e. with dropout active, as dqn.py leaves them) -- its loss and the checksums of every
    policy parameter after the step.  dqn_glue.td_step on the same seeded networks and the same batch in the sampled
    order reproduces them: same loss, same parameters (the same torch ops in the same order, dropout masks included)."""
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    g = golden("qnet.npz")
    torch.manual_seed(int(g["td_seed"]))
    policy, target = glue.QNet(6), glue.QNet(6)           # dqn.py:14-16: policy, target, target <- policy
    target.load_state_dict(policy.state_dict())
    opt = torch.optim.Adam(policy.parameters(), float(g["td_lr"]))   # dqn.py:19
    order = torch.from_numpy(g["td_order"])
    f = lambda k: torch.from_numpy(g[k]).float()[order]   # noqa: E731
    batch = {"s0": f("td_s0"), "a0": f("td_a0"), "s1": f("td_s1"), "a1": f("td_a1"),
             "reward": torch.from_numpy(g["td_rew"])[order], "done": torch.from_numpy(g["td_done_b"])[order]}
    loss = float(glue.td_step(policy, target, opt, batch, gamma=float(g["gamma"])))
    assert abs(loss - float(g["td_loss"])) <= 1e-4 * abs(float(g["td_loss"])), (loss, float(g["td_loss"]))
    sd = policy.state_dict()
    names = [str(x) for x in g["td_after_names"]]
    assert names == sorted(sd)
    sums = np.array([[float(sd[k].double().sum()), float((sd[k].double() ** 2).sum())] for k in names])
    assert np.allclose(sums, g["td_after_sums"], rtol=1e-6, atol=1e-6), np.abs(sums - g["td_after_sums"]).max()


def test_epsilon_schedule_and_checkpoint_dirs_as_the_reference_computes_them(golden):
    """Fixture G9/G10 extras: DQNFirst.update_epsilon (dqn.py:73-76) at six episode counts and config.name_dir
    (config.py:30-31) on five names, evaluated by the reference's own code in the build container."""
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    metrics = importlib.import_module("doudizhu-rl_amd.metrics")
    g = golden("qnet.npz")
    for e, v in zip(g["eps_episodes"], g["eps_values"]):
        assert abs(glue.epsilon_schedule(int(e)) - float(v)) < 1e-12
    assert [float(x) for x in g["hyper"]] == [glue.EPSILON_HIGH, glue.EPSILON_LOW, 20000.0, 256.0, float(glue.DECAY), 20.0]
    for n, d2, d1 in zip(g["name_dir_in"], g["name_dir_out"], g["name_dir_out_split1"]):
        assert metrics.name_dir(str(n)) == str(d2) and metrics.name_dir(str(n), 1) == str(d1)


def test_shared_row_key_determines_the_face_column(oracle):
    """The key of the shared-rows Q forward (csrc/ddz_qnet.h section 5, k_qs_mark): (rank, hand_r, taken_r, b1_r, b2_r) and the
    pair (n1, n2) reduced by its gcd -- 0 where hand_r + taken_r >= total (no prob slot is set) -- must DETERMINE the face
    column of EnvCooperationSimplify (envi.py:201-217): on random mid-game states of the oracle env, all (table, rank)
    instances with one key have bit-identical columns of the oracle's `face` (so one first-layer / fc1 row serves them all)."""
    T = 4096
    env = oracle.OracleEnv(T, seed=77)
    env.reset()
    seen = {}
    for rounds in (0, 9, 23, 41, 66):
        if rounds:
            env.rollout_random(rounds)
        face = env.observe(3)                                            # [T, 6, 15, 4]
        cols = np.ascontiguousarray(face.transpose(0, 2, 1, 3)).reshape(T, 15, 24)
        st = np.asarray(env.state).reshape(T, 11, 16).astype(np.int64)
        role = st[:, 10, 0]
        ar = np.arange(T)
        rm1, rp1 = (role + 2) % 3, (role + 1) % 3
        hand, taken = st[ar, role, :15], st[:, 9, :15]
        b1, b2 = st[ar, 6 + rm1, :15], st[ar, 6 + rp1, :15]
        n1, n2 = st[ar, rp1, 15], st[ar, rm1, 15]
        g = np.gcd(n1, n2)
        g[g == 0] = 1
        total = np.where(np.arange(15) < 13, 4, 1)[None, :]
        ncode = np.where(hand + taken >= total, 0, ((n1 // g) * 21 + n2 // g)[:, None])
        key = ((((np.arange(15)[None, :] * 5 + hand) * 5 + taken) * 5 + b1) * 5 + b2) * 441 + ncode
        assert key.max() < 15 * 625 * 441
        for k, c in zip(key.reshape(-1).tolist(), cols.reshape(-1, 24)):
            b = c.tobytes()
            assert seen.setdefault(k, b) == b                            # one key, one column -- across tables AND states
    assert len(seen) > 2000                                              # (and the states were varied enough to mean something)
