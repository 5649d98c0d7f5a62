"""GPU (-m gpu): the needed-rows form of the ragged Q forward (csrc/ddz_qnet.h; BASELINE configs[2]) through the C ABI:
the hand-written fp32 MFMA GEMM against exact integer products and an fp64 reference, the need sets / row layout bit for
bit against the torch statement (dqn_glue.FactorisedQ.needed_torch), every stage's values against it and against the
literal nn.Conv2d network (net.py:81-102; fp32, tolerance 1e-5 absolute on outputs of magnitude ~0.1: summation order
only), and the whole PolicyLoop captured in a hipGraph."""
import copy
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("doudizhu-rl_amd")


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("M,K", [(256, 32), (300, 256), (1000, 3840), (257, 64), (5, 32)])
def test_fc1_dense_is_an_exact_f32_gemm(pkg, M, K):
    """ddz_q_fc1_dense (k_fc1: v_mfma_f32_32x32x2_f32).  Integer-valued operands whose products and sums are exact in fp32:
    the result must EQUAL the integer product (an asymmetric B, rows beyond the last full tile, C added in); random fp32
    operands: every element within 2e-6 * sum |a b| of the fp64 product (a k-ordered fmaf chain)."""
    g = torch.Generator(device="cpu").manual_seed(M * 7 + K)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (K, 256), generator=g).float()
    w[:, 1::3] += torch.arange(K).float()[:, None] % 5                     # asymmetric, column-dependent
    c0 = torch.randint(-50, 50, (M, 256), generator=g).float()
    c = c0.clone().to(_dev())
    pkg.q_fc1_dense(a.to(_dev()), w.to(_dev()), c)
    want = c0.double() + a.double() @ w.double()
    assert float(want.abs().max()) < 2 ** 23
    assert torch.equal(c.cpu().double(), want)
    a = torch.randn((M, K), generator=g)
    w = torch.randn((K, 256), generator=g)
    c = torch.zeros((M, 256), device=_dev())
    pkg.q_fc1_dense(a.to(_dev()), w.to(_dev()), c)
    want = a.double() @ w.double()
    bound = 2e-6 * (a.abs().double() @ w.abs().double()) + 1e-30
    assert bool(((c.cpu().double() - want).abs() <= bound).all())
    with pytest.raises(pkg.DdzError):
        pkg.q_fc1_dense(torch.zeros((M, 24), device=_dev()), torch.zeros((24, 256), device=_dev()), c)   # K not a multiple of the chunk


def _csr_of_slab(env):
    off, rows, _ = env.slab_to_csr(rows_per_table=512)
    n = int(off[-1])
    return off.clone(), rows[:max(n, 1)].clone(), n


@pytest.mark.parametrize("P,variant", [(6, 3), (4, 0), (9, 2)])
def test_needed_rows_kernels_vs_torch_statement_and_literal_network(pkg, P, variant):
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    T = 700                                                  # not a multiple of the 16 / 128 / 256 tile sizes
    torch.manual_seed(3 + P)
    net = glue.QNet(P).to(_dev()).eval()
    net_cpu = copy.deepcopy(net).cpu()
    env = pkg.BatchedEnv(T, seed=11 + P, device=_dev())
    env.reset()
    fq = glue.FactorisedQ(net)
    fqc = glue.FactorisedQ(net_cpu)
    for rounds in (0, 9, 23):                                # fresh deals (20-card leads), then mixed states
        env.rollout_random(rounds) if rounds else None
        env.legal_slab()
        face = env.observe(variant)
        nu = fq.needed(env, face, gemm="mfma")
        w = fq._ws[("needed", face.device, T)]
        off, rows, n = _csr_of_slab(env)
        ref = fqc.needed_torch(face.cpu(), rows.cpu(), off.cpu())
        # layout: bit for bit
        assert torch.equal(nu.row_index.cpu(), ref.row_index)
        assert nu.seg.cpu()[:34].tolist() == ref.seg[:34].tolist() and int(nu.seg[33]) == 0
        used = int(ref.seg[15])
        needed = ref.row_index >= 0
        rows_used = ref.row_index[needed].long()
        # values: first layer, the two GEMMs, the row stage
        assert float((w["y0"].cpu() - ref.y0).abs().max()) < 2e-6
        assert float((w["dy"].cpu()[rows_used] - ref.dy[rows_used]).abs().max()) < 2e-6
        assert float((nu.h0.cpu() - ref.h0).abs().max()) < 1e-5
        assert float((nu.d.cpu()[rows_used] - ref.d[rows_used]).abs().max()) < 1e-5
        q = fq.q_slab(env, nu)
        counts = env.counts.long()
        valid = torch.arange(env.slab_stride, device=_dev())[None, :] < counts[:, None]
        q_csr = q[valid].cpu()                                # slab order == CSR order
        qt = fqc.q_csr_needed(ref, rows.cpu(), off.cpu())[:n]
        assert q_csr.numel() == n and float((q_csr - qt).abs().max()) < 1e-5
        seg_t = torch.repeat_interleave(torch.arange(T), counts.cpu())
        pick = torch.arange(0, n, 5)
        acts = (rows.cpu()[pick, :15].float()[:, :, None] > torch.arange(4)[None, None, :]).float()
        with torch.no_grad():
            want = net_cpu(face.cpu()[seg_t[pick]], acts)[:, 0]
        assert float((q_csr[pick] - want).abs().max()) < 1e-5
        assert used % glue.fc_tile() == 0 and used <= w["cap"]
        # the dense GEMM by the library gives the same H0 up to summation order
        nu2 = fq.needed(env, face, gemm="torch")
        assert float((nu2.h0.cpu() - ref.h0).abs().max()) < 1e-5
    assert env.status() == 0
    # a row_index that does not belong to the lists: nothing is dereferenced beyond the buffers, status bit 5
    wild = nu.row_index.clone()
    wild[wild >= 0] += w["cap"]
    qw = env.q_slab_needed(nu.h0, nu.d, wild, fq.w2, fq.b2)
    assert bool(torch.isfinite(qw[valid]).all()) and env.status() & 32


@pytest.mark.parametrize("T", [700, 37, 5000])
def test_shared_rows_form_of_h0_equals_the_dense_form(pkg, T):
    """FactorisedQ.needed(shared=True) (csrc/ddz_qnet.h section 5: one row per distinct (rank, face column), no K = 3840
    GEMM) against the dense form and the torch statement on fresh deals and mixed states: the row layout (every (t, r)
    points at a row of ITS rank's segment whose representative has bit for bit the same face column; distinct columns have
    distinct rows; padding rows are zero), H0 within 1e-5 (fp32 summation order), D bit for bit (the same kernel, y0 = null),
    q within 1e-5 of the dense form and of the literal network."""
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    torch.manual_seed(17)
    net = glue.QNet(6).to(_dev()).eval()
    net_cpu = copy.deepcopy(net).cpu()
    env = pkg.BatchedEnv(T, seed=29, device=_dev())
    env.reset()
    fq, fq2, fq3 = glue.FactorisedQ(net), glue.FactorisedQ(net), glue.FactorisedQ(net)
    for rounds in (0, 7, 30, 61):
        env.rollout_random(rounds) if rounds else None
        env.legal_slab()
        face = env.observe(3)
        dense = fq2.needed(env, face, gemm="torch")
        nu = fq.needed(env, face, shared=True)
        w = fq._ws[("needed", face.device, T)]
        assert w["y0"] is None                                            # the [T, 3840] operand does not exist in this form
        rows, rep, seg = w["srows"].cpu().long(), w["srep"].cpu().long(), w["sseg"].cpu().tolist()
        assert seg[33] == 0 and seg[15] % glue.fc_tile() == 0 and seg[15] <= w["scap"]
        assert bool((rows[:, 15] == -1).all())
        cols = face.cpu().permute(0, 2, 1, 3).reshape(T * 15, 6, 4)       # column of instance 15 t + r
        r15 = rows[:, :15]
        for r in range(15):
            lo, hi = seg[r], (seg[r + 1] if r < 14 else seg[15])
            assert bool(((r15[:, r] >= lo) & (r15[:, r] < hi)).all())
        inst = rep[r15]                                                   # [T,15] representative 16 t' + r'
        assert bool((inst >= 0).all()) and bool(((inst & 15) == torch.arange(15)[None, :]).all())
        rep_cols = cols[(inst >> 4) * 15 + (inst & 15)]
        assert torch.equal(rep_cols, cols.view(T, 15, 6, 4))              # same column, bit for bit
        # one row per distinct KEY (hand_r, taken_r, b1_r, b2_r, canonical (n1, n2)) of a rank, numbered in key order; keys are at
        # least as fine as columns, never coarser (checked above: the representative's column is the instance's, bit for bit)
        st = env.state.cpu().view(T, 11, 16).long()
        role = st[:, 10, 0]
        ar = torch.arange(T)
        rm1, rp1 = (role + 2) % 3, (role + 1) % 3
        n1, n2 = st[ar, rp1, 15], st[ar, rm1, 15]
        gg = torch.gcd(n1, n2).clamp(min=1)                                  # canonical (n1, n2): n / (n1 + n2) is what the planes hold
        total = torch.where(torch.arange(15) < 13, 4, 1)[None, :]
        ncode = torch.where(st[ar, role, :15] + st[:, 9, :15] >= total, torch.zeros(1, dtype=torch.long),
                            ((n1 // gg) * 21 + n2 // gg)[:, None])           # ... and only where known < total
        key = ((((st[ar, role, :15] * 5 + st[:, 9, :15]) * 5 + st[ar, 6 + rm1, :15]) * 5 + st[ar, 6 + rp1, :15]) * 441 + ncode)
        n_rows = 0
        for r in range(15):
            uk, inv = torch.unique(key[:, r], return_inverse=True)          # sorted: the row order inside the segment
            assert torch.equal(r15[:, r], seg[r] + inv)
            assert torch.unique(cols.view(T, 15, 24)[:, r], dim=0).shape[0] <= uk.numel()
            n_rows += uk.numel()
        assert seg[32] == n_rows
        used = torch.zeros(w["scap"], dtype=torch.bool)
        used[r15.reshape(-1)] = True
        assert bool((rep[~used] == -1).all()) and float(w["ys"].cpu()[: seg[15]][~used[: seg[15]]].abs().max() if (~used[: seg[15]]).any() else 0.0) == 0.0
        # values
        assert float((nu.h0 - dense.h0).abs().max()) < 1e-5
        ri = nu.row_index.cpu()
        sel = ri[ri >= 0].long()
        assert torch.equal(nu.row_index, dense.row_index) and torch.equal(nu.d.cpu()[sel], dense.d.cpu()[sel])
        q = fq.q_slab(env, nu).clone()
        q2 = fq2.q_slab(env, dense)
        counts = env.counts.long()
        valid = torch.arange(env.slab_stride, device=_dev())[None, :] < counts[:, None]
        assert float((q[valid] - q2[valid]).abs().max()) < 1e-5
        # the needed rows shared as well (section 6): one D row per distinct (shared row, count); the same D values bit for bit
        # (the same expression per row, a k-ordered chain per row whatever its tile), hence bit-identical q
        na = fq3.needed(env, face, shared="all")
        w3 = fq3._ws[("needed", face.device, T)]
        ri2, dseg, drep = na.row_index.cpu().long(), w3["dseg"].cpu().tolist(), w3["drep"].cpu().long()
        assert torch.equal(ri2 >= 0, ri >= 0) and dseg[33] == 0 and dseg[15] % glue.fc_tile() == 0 and dseg[15] <= w3["cap"]
        assert torch.equal(w3["srows"].cpu().long(), rows) and float((na.h0 - nu.h0).abs().max()) == 0.0
        col = torch.arange(64)[None, :].expand(T, 64)
        rk = torch.where(col < 52, col // 4, 13 + (col - 52).clamp(min=0))
        cc = torch.where(col < 52, col % 4 + 1, torch.ones_like(col))
        need = ri >= 0
        slot = (rows.gather(1, rk.clamp(max=14)) * 4 + cc - 1)[need]       # 4 * shared row + c - 1 of every needed triple
        assert torch.equal(drep[ri2[need]], slot)                          # its D row is the row of that slot ...
        uq = torch.unique(slot)
        assert dseg[32] == uq.numel()                                      # ... one row per distinct slot, in slot order per rank
        assert torch.equal(torch.unique(ri2[need]), torch.sort(ri2[need].unique()).values) and torch.unique(ri2[need]).numel() == uq.numel()
        assert torch.equal(w3["drow_cnt"].cpu().long()[ri2[need]], cc[need])
        for r in range(15):
            lo, hi = dseg[r], (dseg[r + 1] if r < 14 else dseg[15])
            m = need & (rk == r)
            assert bool(((ri2[m] >= lo) & (ri2[m] < hi)).all())
        assert torch.equal(na.d.cpu()[ri2[need]], nu.d.cpu()[ri[need].long()])
        q3 = fq3.q_slab(env, na)
        assert torch.equal(q3[valid], q[valid])
        off, lrows, n = _csr_of_slab(env)
        seg_t = torch.repeat_interleave(torch.arange(T), counts.cpu())
        pick = torch.arange(0, n, 7)
        acts = (lrows.cpu()[pick, :15].float()[:, :, None] > torch.arange(4)[None, None, :]).float()
        with torch.no_grad():
            want = net_cpu(face.cpu()[seg_t[pick]], acts)[:, 0]
        assert float((q[valid].cpu()[pick] - want).abs().max()) < 1e-5
    assert env.status() == 0
    # argument errors: a capacity that could overflow, a face with other planes
    engine = importlib.import_module("doudizhu-rl_amd.engine")
    with pytest.raises(pkg.DdzError):
        env.q_shared_rows(w["sws"], glue.fc_tile() * 15, w["srows"], w["srep"][: glue.fc_tile() * 15], w["sseg"])
    with pytest.raises(ValueError):
        glue.FactorisedQ(glue.QNet(4).to(_dev()).eval()).needed(env, env.observe(0), shared=True)
    assert engine is not None


def test_need_capacity_overflow_is_flagged_not_written(pkg):
    """row_capacity too small for the needed rows: status bit 1, seg[33] = 1, the rows that do not fit are -1 (never an
    index beyond the capacity); a capacity that is no multiple of the fc1 tile is an argument error."""
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    engine = importlib.import_module("doudizhu-rl_amd.engine")
    T = 512
    env = pkg.BatchedEnv(T, seed=5, device=_dev())
    env.reset(); env.legal_slab()                            # 20-card leads: ~20 needed rows per table
    cap = 15 * glue.fc_tile() + 1024
    row_index = torch.full((T, 64), -1, dtype=torch.int32, device=_dev())
    seg = torch.zeros(40, dtype=torch.int32, device=_dev())
    scratch = torch.zeros(engine.q_need_scratch_bytes(T), dtype=torch.uint8, device=_dev())
    row_cnt = torch.zeros(cap + 64, dtype=torch.uint8, device=_dev())
    env.q_need(cap, scratch, row_index, seg, row_cnt)
    s = seg.cpu().tolist()
    assert s[33] == 1 and s[15] <= cap and s[32] > cap and int(row_index.max()) < cap
    assert env.status() & 2
    # the count of every row that exists, nothing behind the capacity
    idx = row_index.cpu()
    cnt = row_cnt.cpu()
    col = torch.arange(64)
    want_c = torch.where(col < 52, col % 4 + 1, torch.ones_like(col))
    sel = idx >= 0
    assert bool((cnt[idx[sel].long()] == want_c[None, :].expand_as(idx)[sel].to(torch.uint8)).all()) and int(cnt[cap:].max()) == 0
    with pytest.raises(pkg.DdzError):
        env.q_need(cap + 5, scratch, row_index, seg, row_cnt)
    assert glue is not None


def test_policy_loop_needed_form_is_graph_capturable(pkg):
    """PolicyLoop.step in the needed form has no host synchronisation (no .cpu(), no size-dependent shape): 6 iterations
    captured in a hipGraph and replayed 3 times == the same 18 iterations issued one by one (states, faces, choices and
    q values bit for bit), with either implementation of the dense GEMM."""
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    for gemm, shared in (("mfma", False), ("torch", False), ("torch", True), ("torch", "all")):
        T, K = 1500, 6
        torch.manual_seed(1)
        net = glue.QNet(6).to(_dev()).eval()
        a = pkg.BatchedEnv(T, seed=21, device=_dev())
        b = pkg.BatchedEnv(T, seed=21, device=_dev())
        a.reset(); b.reset()
        la = glue.PolicyLoop(a, net, face_variant=3, epsilon=0.1, gemm=gemm, shared=shared)
        lb = glue.PolicyLoop(b, net, face_variant=3, epsilon=0.1, gemm=gemm, shared=shared)
        la.run(2); lb.run(2)                                 # workspaces allocated, libraries warm
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                la.run(K)
        torch.cuda.current_stream().wait_stream(s)
        for _ in range(3):
            g.replay()
            lb.run(K)
        torch.cuda.synchronize()
        assert torch.equal(a.state, b.state) and torch.equal(la.face, lb.face) and torch.equal(la.choice, lb.choice)
        valid = torch.arange(a.slab_stride, device=_dev())[None, :] < a.counts.long()[:, None]
        assert torch.equal(a.counts, b.counts) and torch.equal(la.q[valid], lb.q[valid])
        assert a.status() == 0 and a.stats() == b.stats()
