"""Batched transition assembly for the DQN loop of the reference.

The reference closes the transition of role X when X's next observation appears, or at the
terminal ply (game.py:109-167):
  * X acts in state s0 with action a0 (game.py:95-104);
  * the next time X is to move, feedback(X, done=False) stores (s0, a0, 0, s1 = face now,
    a1 = greedy action now, False) (game.py:109-127, called from :131-133, :146-148, :158-160);
  * when any role empties its hand every role with a pending (s0, a0) gets
    (s0, a0, +/-reward_dict[role], s1 = face after the terminal ply, a1 = zeros[15,4], True)
    (game.py:113-123, :134-141, :149-155, :161-167): winners +reward, losers -reward; lord and
    farmers are opposite sides, the two farmers win and lose together; the calls come in play
    order starting behind the winner (lord wins: down, up, lord; down wins: up, lord, down; up
    wins: lord, down, up).
Here that bookkeeping is done for T tables at once with tensor ops (any device).  Pinned by fixture
G11 (tests/golden/gen_game.py: every perceive() call of the reference's own Game.play).

One documented choice: the reference never clears `*_s0 / *_a0` between episodes
(game.py:39-40,132-137; SURVEY.md appendix A "quirks"), so from the second episode of a Game
object on, the first feedback of `down` and of `up` pushes (s0, a0 of the PREVIOUS episode's last
move, 0, s1 = first observation of the new episode, a1, False) -- a transition across a reshuffle;
the lord's stale pair is overwritten unseen.  Default here: pending slots are cleared at the
terminal ply (no such transition); replicate_reference_quirk=True reproduces the reference call
for call (both modes are checked against G11).

Usage per lock-step iteration (all tensors [T, ...]):
    closed = asm.before_step(role, face, chosen_onehot, greedy_onehot)
    ... env.step(auto_reset=False) ...; terminal_face = env.observe(variant)
    ended = asm.after_step(role, done, r, terminal_face); env.reset(mask=done)
Both calls return a dict of the transitions they closed (s0, a0, reward, s1, a1, done,
table, role) ready to append to a replay buffer; within a table the rows are in the reference's
call order.
"""
import torch

REWARD_DICT = {"up": 50.0, "lord": 100.0, "down": 50.0}  # game.py:13-14; role ids 0 up, 1 lord, 2 down


class TransitionAssembler:
    def __init__(self, n_tables, planes, device, reward_dict=None, trained_roles=(True, True, True),
                 replicate_reference_quirk=False):
        """trained_roles: (up, lord, down) -- the roles whose agent keeps training (Game's train_dict, game.py:15-16,
        :95-104,:112): only they open and close transitions.  replicate_reference_quirk: see the module docstring."""
        rd = dict(REWARD_DICT if reward_dict is None else reward_dict)
        self.T, self.P, self.device = int(n_tables), int(planes), torch.device(device)
        self.reward = torch.tensor([rd["up"], rd["lord"], rd["down"]], dtype=torch.float32, device=self.device)
        self.trained = torch.tensor([bool(x) for x in trained_roles], dtype=torch.bool, device=self.device)
        self.quirk = bool(replicate_reference_quirk)
        self.s0 = torch.zeros((self.T, 3, self.P, 15, 4), dtype=torch.float32, device=self.device)
        self.a0 = torch.zeros((self.T, 3, 15, 4), dtype=torch.float32, device=self.device)
        self.pending = torch.zeros((self.T, 3), dtype=torch.bool, device=self.device)
        self.fresh = torch.ones(self.T, dtype=torch.bool, device=self.device)   # no ply of the episode played yet

    @staticmethod
    def _pack(s0, a0, reward, s1, a1, done, table, role):
        return {"s0": s0, "a0": a0, "reward": reward, "s1": s1, "a1": a1, "done": done, "table": table,
                "role": role}

    def before_step(self, role, face, chosen, greedy, active=None):
        """role int[T] actor of each table, face f32[T,P,15,4] its observation, chosen / greedy
        f32[T,15,4] the action it is about to play and the greedy action (a1 of the closing
        transition, game.py:125).  Closes the actor's previous transition, opens a new one.
        active bool[T]: tables that move in this iteration (default all)."""
        role = role.to(self.device).long()
        ar = torch.arange(self.T, device=self.device)
        if active is None:
            active = torch.ones(self.T, dtype=torch.bool, device=self.device)
        active = active.to(self.device).bool()
        act_tr = active & self.trained[role]
        # the first ply of an episode (the lord's) has no feedback in front of it (game.py:129-130): whatever the slot
        # still holds is overwritten unseen
        close = self.pending[ar, role] & act_tr & ~self.fresh
        idx = close.nonzero(as_tuple=True)[0]
        r_idx = role[idx]
        out = self._pack(self.s0[idx, r_idx].clone(), self.a0[idx, r_idx].clone(),
                         torch.zeros(idx.numel(), dtype=torch.float32, device=self.device),
                         face[idx].clone(), greedy[idx].clone(),
                         torch.zeros(idx.numel(), dtype=torch.bool, device=self.device), idx, r_idx)
        act = act_tr.nonzero(as_tuple=True)[0]
        self.s0[act, role[act]] = face[act]
        self.a0[act, role[act]] = chosen[act]
        self.pending[act, role[act]] = True
        self.fresh &= ~active
        return out

    def after_step(self, role, done, r, terminal_face):
        """role int[T] the actor that just moved, done u8[T], r i8[T] (-1 lord won, +1 farmers
        won; rule_play.py:14), terminal_face f32[T,P,15,4] = env.observe() after the ply.
        Closes every pending transition of the finished tables."""
        done = done.to(self.device).bool()
        role = role.to(self.device).long()
        t_idx, r_idx = (self.pending & done[:, None]).nonzero(as_tuple=True)
        # the reference's call order: play order (lord, down, up) starting behind the winner = the actor of this ply
        order = torch.argsort(t_idx * 3 + (r_idx - role[t_idx] - 1) % 3)
        t_idx, r_idx = t_idx[order], r_idx[order]
        lord_won = (r.to(self.device)[t_idx] < 0)
        is_lord = r_idx == 1
        sign = torch.where(lord_won == is_lord, 1.0, -1.0)
        out = self._pack(self.s0[t_idx, r_idx].clone(), self.a0[t_idx, r_idx].clone(),
                         sign * self.reward[r_idx], terminal_face[t_idx].clone(),
                         torch.zeros((t_idx.numel(), 15, 4), dtype=torch.float32, device=self.device),
                         torch.ones(t_idx.numel(), dtype=torch.bool, device=self.device), t_idx, r_idx)
        if not self.quirk:
            self.pending[done] = False
        self.fresh |= done
        return out


def td_target(transitions, q_next, gamma=0.95):
    """y = r + (1 - done) * gamma * Q_target(s1, a1)  (dqn.py:40-41, config.py:8 GAMMA)."""
    return transitions["reward"] + (~transitions["done"]).float() * gamma * q_next.view(-1)


# ------------------------------------------------------------------------------------------------
# The ragged Q forward of the reference's DQN (net.py:81-102, dqn.py:50-71, game.py:95-104) for T tables at once.
#
# The reference evaluates Q(face, action) for EVERY legal action of a state by repeating `face` A times and
# concatenating the action as one more input plane (net.py:87-90): sum_A rows x (C x 15 x 4) inputs per iteration.
# Here the first layer is evaluated factorised.  Its five convolutions are linear in their input and look at ONE
# rank (conv1..4: a (1,k) window, stride 4, on a width-4 input -> one column per rank, net.py:141-144) or ONE
# thermometer slot (conv_shunzi (15,1), net.py:146), and an action plane is a thermometer of its count vector
# (envi.py:139-146), so for a rank r that an action takes `cnt` cards of:
#     maxpool_k conv_k(face + action)[c, r] = max_k ( S_k[t, r, c] + A_k[cnt, c] )          =: Y[t, r, cnt, c]
# with S = the face part (one GEMM per table, NOT per legal row) and A a 5 x 4 x 256 table of the action-plane
# weights -- the same for every rank, and cnt = 0 for every rank the action does not touch.  fc1 is linear, so its
# pre-activation is a sum over the 15 ranks of U[r, t, cnt_r, :] = fc1_r @ Y[t, r, cnt_r, :] (+ the conv_shunzi
# branch, linear end to end, folded in: a per-table vector and a per-(rank, count) vector).  Per iteration:
#     tables(face)  -> U [15, 5, T, 256]   dense, fixed shapes, plain torch GEMMs (hipBLASLt) -- no ragged dimension
#     per legal row -> q = fc2(relu(sum_r U[r, cnt_r, t] + Z[r, cnt_r]))   a gather-sum + a 256-dot per row
# The per-row stage runs over the slab lists in the engine (ddz_q_slab: no CSR, no host sync, no padded rows), or
# over CSR rows with plain torch ops (q_csr: the reference statement, used by the tests).
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

_CONV_CH = 256
_FC_TILE = None


def fc_tile():
    """rows per tile of the engine's fc1 kernel (ddz_q_fc1_tile_rows, csrc/ddz_qnet.h FC_M): the rank segments of the needed
    rows start at multiples of it"""
    global _FC_TILE
    if _FC_TILE is None:
        from . import _lib
        _FC_TILE = int(_lib.lib().ddz_q_fc1_tile_rows())
    return _FC_TILE


class QNet(nn.Module):
    """The reference's Q-network family (net.py:66-150: NetComplicated 5 input planes, NetMoreComplicated 8,
    NetCooperation 10, NetCooperationSimplify 7), same parameter names and shapes (state_dict-compatible); forward is
    the literal evaluation of net.py:81-102.  `planes` = planes of `face` (4 / 7 / 9 / 6); + 1 for the action."""

    def __init__(self, planes=6):
        super().__init__()
        c = int(planes) + 1
        self.planes = int(planes)
        self.conv1 = nn.Conv2d(c, _CONV_CH, (1, 1), (1, 4))
        self.conv2 = nn.Conv2d(c, _CONV_CH, (1, 2), (1, 4))
        self.conv3 = nn.Conv2d(c, _CONV_CH, (1, 3), (1, 4))
        self.conv4 = nn.Conv2d(c, _CONV_CH, (1, 4), (1, 4))
        self.conv_shunzi = nn.Conv2d(c, _CONV_CH, (15, 1), 1)
        self.pool = nn.MaxPool2d((1, 4))
        self.drop = nn.Dropout(0.5)
        self.fc1 = nn.Linear(_CONV_CH * (15 + 4), 256)
        self.fc2 = nn.Linear(256, 1)

    def forward(self, face, actions):
        """face [P,15,4] or [n,P,15,4], actions [n,15,4] -> Q [n,1]  (net.py:81-102)"""
        if face.dim() == 3:
            face = face.unsqueeze(0).repeat((actions.shape[0], 1, 1, 1))
        x = torch.cat((face, actions.unsqueeze(1)), dim=1)
        y = torch.cat([f(x) for f in (self.conv1, self.conv2, self.conv3, self.conv4)], -1)
        y = self.pool(y).view(actions.shape[0], -1)
        z = self.conv_shunzi(x).view(actions.shape[0], -1)
        h = self.drop(torch.cat([y, z], -1))
        return self.fc2(F.relu(self.fc1(h)))


class FactorisedQ:
    """Inference form of a QNet: the weight-only tables of the factorisation above, cached until the weights change
    (refresh() is called automatically when a parameter's version counter moved).  Eval semantics (no dropout)."""
    def __init__(self, net, chunk_tables=16384):
        self.net, self.chunk = net, int(chunk_tables)
        self.two_streams = True        # needed(shared="all"): the D chain on a side stream beside the H0 chain
        # packed form: fifteen fc1 GEMMs over exactly the rows that exist (default), or ONE batched GEMM over segments
        # padded to the longest (True).  Measured at 65,536 tables: steady state (thousands of iterations in) 4.00 against
        # 4.23 ms per iteration; a batch of young games (30 iterations in, 16 % padding) 3.75 against 3.60 ms.
        self.batched_gemm = False
        self.P = net.planes
        self._ver = None
        self._ws = {}
        self.refresh()

    def _versions(self):
        # (id and storage address too: load_state_dict(assign=True) / a swapped Parameter can carry an equal version)
        return tuple((id(p), p.data_ptr(), p._version) for p in self.net.parameters()) + (next(self.net.parameters()).device,)

    @torch.no_grad()
    def refresh(self):
        n, P, H = self.net, self.P, _CONV_CH
        dev, dt = n.fc1.weight.device, torch.float32
        C = P + 1
        convs = (n.conv1, n.conv2, n.conv3, n.conv4)
        Wf = torch.zeros((P, 4, 4, H), dtype=dt, device=dev)           # [plane, slot j, conv k, channel]
        A = torch.zeros((5, 4, H), dtype=dt, device=dev)               # [count, conv k, channel]
        for k, cv in enumerate(convs):
            w = cv.weight[:, :, 0, :]                                  # [H, C, k+1]
            Wf[:, : k + 1, k, :] = w[:, :P, :].permute(1, 2, 0)
            for cnt in range(1, 5):
                A[cnt, k] = w[:, C - 1, : min(k + 1, cnt)].sum(dim=1)
        self.Wf = Wf.reshape(P * 4, 4 * H).contiguous()
        self.bias_f = torch.cat([cv.bias for cv in convs]).contiguous()
        self.A = A.contiguous()
        H1 = n.fc1.out_features
        W1 = n.fc1.weight
        W1y = W1[:, : 15 * H].reshape(H1, H, 15)                       # input index c * 15 + r (net.py:94 view)
        W1z = W1[:, 15 * H:].reshape(H1, H, 4)                         # input index c * 4 + w  (net.py:96)
        Ws = n.conv_shunzi.weight[:, :, :, 0]                          # [H, C, 15]
        Mz = torch.einsum("ocw,cpr->prwo", W1z, Ws)                    # [C, 15, 4, H1]: conv_shunzi then fc1, composed
        self.Mz_f = Mz[:P].reshape(P * 60, H1).contiguous()            # face part: one GEMM per table
        Z = torch.zeros((15, 5, H1), dtype=dt, device=dev)             # action part per (rank, count): weights only
        Z[:, 1:] = Mz[C - 1].cumsum(dim=1)
        self.Z = Z.contiguous()
        self.base = (n.fc1.bias + torch.einsum("ocw,c->o", W1z, n.conv_shunzi.bias)).contiguous()
        W2 = W1y.permute(2, 1, 0).contiguous()                         # [r, c, o]: fc1 per rank
        self.W2 = W2
        # fc1 per rank with the table term's rows appended (the shared-rows form, csrc/ddz_qnet.h section 5): a shared row
        # carries its 24 column values behind its 256 first-layer values, so [fc1_r ; Mz[:, r] ; 0] (K = 288) gives
        # Y x fc1_r + column x Mz_r in one product
        self.W2x = torch.cat([W2, Mz[:P].permute(1, 0, 2, 3).reshape(15, P * 4, H1),
                              torch.zeros((15, 32 - P * 4, H1), dtype=dt, device=dev)], dim=1).contiguous() if P * 4 <= 32 else None
        self.Wd = W2.reshape(15 * H, H1)                               # the dense GEMM's right operand: K = 15 * 256, rank-major
        # one GEMM batch per (rank, count): ranks 3..2 have counts 0..4 (65 batches), the two jokers counts 0..1
        self.W2_main = W2[:13, None].expand(13, 5, H, H1).reshape(65, H, H1).contiguous()
        self.W2_jok = [W2[r, None].expand(2, H, H1).contiguous() for r in (13, 14)]
        self.w2 = n.fc2.weight[0].contiguous()
        self.b2 = n.fc2.bias.detach().clone()
        self.H, self.H1 = H, H1
        self._ver = self._versions()
        self._ws = {}

    def _workspace(self, Tc, dev, fused):
        key = (Tc, dev, fused)
        if key not in self._ws:
            if len(self._ws) > 3:
                self._ws.clear()
            Y = torch.zeros((15, 5, Tc, self.H), dtype=torch.float32, device=dev)
            S = tmp = None
            if not fused:
                S = torch.empty((15 * Tc, 4, self.H), dtype=torch.float32, device=dev)
                tmp = torch.empty((15 * Tc, 4, self.H), dtype=torch.float32, device=dev)
            self._ws[key] = (Y, S, tmp)
        return self._ws[key]

    @torch.no_grad()
    def tables(self, face, out=None, fused=None):
        """face f32 [T,P,15,4] -> U f32 [15,5,T,H1]: fc1's pre-activation contribution of rank r when the action takes
        cnt cards of it (the per-table terms -- fc1 bias, the face part of conv_shunzi -- ride on rank 0; counts 2..4 of
        the two joker ranks are never written: pass a zero-initialised `out`).  Fixed shapes, no host sync; tables are
        processed in chunks to bound the workspace.
        fused (default: on a GPU): Y = max-pooled first layer per (rank, count, table) comes from the engine's
        ddz_q_features in one pass over `face`; fused=False is the same stage in plain torch ops (the statement the
        kernel is tested against; it reads and writes the [T,15,4,256] conv output ten times)."""
        if self._ver != self._versions():
            self.refresh()
        T, P, H, H1 = face.shape[0], self.P, self.H, self.H1
        if tuple(face.shape[1:]) != (P, 15, 4):
            raise ValueError(f"face must be [T,{P},15,4]")
        if fused is None:
            fused = face.is_cuda
        U = out if out is not None else torch.zeros((15, 5, T, H1), dtype=torch.float32, device=face.device)
        if tuple(U.shape) != (15, 5, T, H1) or not U.is_contiguous():
            raise ValueError("out must be a contiguous [15,5,T,256] tensor")
        U75 = U.view(75, T, H1)
        for t0 in range(0, T, self.chunk):
            t1 = min(T, t0 + self.chunk)
            Tc = t1 - t0
            f = face[t0:t1]
            Y, S, tmp = self._workspace(Tc, face.device, fused)
            if fused:
                from .engine import q_features
                q_features(f.contiguous(), self.Wf, self.bias_f, self.A, Y)
            else:
                X = f.permute(2, 0, 1, 3).reshape(15 * Tc, P * 4)      # rank-major rows: (r, t) x (plane, slot)
                torch.addmm(self.bias_f, X, self.Wf, out=S.view(15 * Tc, 4 * H))
                for cnt in range(5):
                    torch.add(S, self.A[cnt], out=tmp)
                    Y[:, cnt] = tmp.amax(dim=1).view(15, Tc, H)        # max over the four convs = the (1,4) max-pool
            Y75 = Y.view(75, Tc, H)
            torch.bmm(Y75[:65], self.W2_main, out=U75[:65, t0:t1])     # ranks 3..2, counts 0..4
            for k, r in enumerate((13, 14)):                           # the jokers: counts 0, 1
                torch.bmm(Y75[5 * r: 5 * r + 2], self.W2_jok[k], out=U75[5 * r: 5 * r + 2, t0:t1])
            U[0, :, t0:t1] += torch.addmm(self.base, f.reshape(Tc, P * 60), self.Mz_f)
        return U

    # ---- packed form: only the (rank, count, table) rows a legal move can use ----
    @torch.no_grad()
    def pack(self, hands):
        """hands int [T,15] (BatchedEnv.actor_hands: what the acting role holds) -> (row_index int32 [T,64], rank_row0):
        the layout of ddz_q_features_packed / ddz_q_slab_packed.  A legal move takes at most hands[t][r] cards of rank r,
        so of the 69 (rank, count) rows of a table only 15 + (cards in hand) are ever read: count 0 of every rank (the
        first T rows of rank r's segment) and counts 1..hands[t][r] (behind them, in table order).  The segments follow
        each other without gaps (fifteen GEMMs of different row counts), or -- batched_gemm -- all have the length of the
        longest, rounded up to 2048 rows (rank_row0[r] = r M; the rows behind a shorter rank's last held count are
        padding): ONE batched GEMM [15, M, 256] x [15, 256, 256].  rank_row0 = 16 python ints (rank r's first row; [15] =
        the number of rows); the sizes come from ONE small device -> host copy, the only sync of the packed forward."""
        T, dev = hands.shape[0], hands.device
        hc = hands.clamp(0, 4)
        hc[:, 13:] = hc[:, 13:].clamp(max=1)                           # a joker exists once
        per_rank = hc.t().contiguous()                                 # [15, T]
        flat = per_rank.view(-1)
        excl = (flat.cumsum(0) - flat).view(15, T)                     # ONE 1-D scan: held rows before (r, t), all ranks
        rel = T + excl - excl[:, :1]                                   # row of (r, t, count 1) inside rank r's segment
        sizes = [int(x) for x in (T + per_rank.sum(1)).cpu()]          # the sync
        if self.batched_gemm:
            # every segment as long as the longest, rounded up to 2048 rows (a coarse grid: the GEMM's shape then repeats
            # from iteration to iteration)
            M = (max(sizes) + 2047) // 2048 * 2048
            host = [r * M for r in range(16)]
        else:                                                          # fifteen GEMMs over exactly the rows that exist
            host = [0] * 16
            for r in range(15):
                host[r + 1] = host[r] + sizes[r]
        base = (rel + torch.tensor(host[:15], device=dev)[:, None]).t()   # [T,15]
        c = torch.arange(4, device=dev)
        idx = torch.where(c < hc[:, :, None], base[:, :, None] + c, -1).to(torch.int32)   # [T,15,4]
        row_index = torch.full((T, 64), -1, dtype=torch.int32, device=dev)
        row_index[:, :52] = idx[:, :13].reshape(T, 52)
        row_index[:, 52], row_index[:, 53] = idx[:, 13, 0], idx[:, 14, 0]
        return row_index, host

    @torch.no_grad()
    def tables_packed(self, face, hands, fused=None):
        """face f32 [T,P,15,4], hands int [T,15] -> PackedU: the rows of tables() a legal move can use (a third of them
        at ~10 cards per hand: a third of the fc1 GEMM and of the first layer's stores), one GEMM per rank over that rank's
        rows (or one batched GEMM over padded segments: batched_gemm), the per-table term on its own.  One host sync (pack).  fused=False (CPU): the same rows
        gathered from the plain-torch tables() -- the statement the packed kernels are tested against.
        The result's .u is THIS object's cached workspace, not a copy: the next tables_packed() call on the same
        FactorisedQ overwrites it (and a batch holding more cards than any before reallocates it) -- consume a PackedU
        before asking for the next one, or clone .u."""
        if self._ver != self._versions():
            self.refresh()
        T, P, H, H1 = face.shape[0], self.P, self.H, self.H1
        if tuple(face.shape[1:]) != (P, 15, 4) or tuple(hands.shape) != (T, 15):
            raise ValueError(f"face must be [T,{P},15,4] and hands [T,15]")
        if fused is None:
            fused = face.is_cuda
        if T * 69 * max(H, H1) >= 1 << 31:
            raise ValueError(f"tables_packed: {T} tables can need more packed rows than ddz_q_features_packed indexes with 32 "
                             "bits; call it on slices of at most 120,000 tables (tables() chunks by itself)")
        row_index, row0 = self.pack(hands)
        n = row0[15]
        key = ("packed", face.device)
        if key not in self._ws or self._ws[key][0].shape[0] < n:
            cap = max(n, 30 * T) * 9 // 8                              # (grows when a batch holds more cards than any before)
            self._ws[key] = (torch.zeros((cap, H), dtype=torch.float32, device=face.device),
                             torch.zeros((cap, H1), dtype=torch.float32, device=face.device))
        Yc, Uc = self._ws[key]
        if fused:
            from .engine import q_features_packed
            q_features_packed(face.contiguous(), self.Wf, self.bias_f, self.A, row_index, row0, Yc)
        else:
            Y = self._first_layer_torch(face)                          # [15,5,T,H]
            for r in range(15):
                Yc[row0[r]: row0[r] + T] = Y[r, 0]
            cols = [(r, c) for r in range(13) for c in range(1, 5)] + [(13, 1), (14, 1)]
            for k, (r, c) in enumerate(cols):
                dst = row_index[:, k].long()
                m = dst >= 0
                Yc[dst[m]] = Y[r, c][m]
        if self.batched_gemm:
            M = row0[1]
            torch.bmm(Yc[:n].view(15, M, H), self.W2, out=Uc[:n].view(15, M, H1))
        else:
            for r in range(15):
                torch.mm(Yc[row0[r]: row0[r + 1]], self.W2[r], out=Uc[row0[r]: row0[r + 1]])
        tab = torch.addmm(self.base, face.reshape(T, P * 60), self.Mz_f)
        return PackedU(Uc, row_index, row0, tab)

    # ---- needed form: H0 per table from ONE dense GEMM + D only for the (rank, count) rows some legal move uses ----
    @torch.no_grad()
    def needed(self, env, face, gemm="torch", shared=False):
        """face f32 [T,P,15,4] of env's CURRENT states (its slab lists are read on the device) -> NeededU: h0 f32 [T,256],
        d f32 [rows,256], row_index int32 [T,64], seg int32 [40] (device).  Nothing crosses to the host; every launch is
        graph-capturable.  The rows GEMM (D = dY x fc1[rank], segment sizes in device memory) is always the engine's fp32
        MFMA kernel (ddz_q_fc1_rows: a library GEMM would need the sizes on the host).  The dense GEMM (a plain
        [T, 3840] x [3840, 256] product) is torch.addmm = hipBLASLt by default (gemm="torch": 148 TFLOP/s in the loop), or
        the same MFMA kernel (gemm="mfma", ddz_q_fc1_dense: 125 TFLOP/s; six geometries measured, tools/fc1_probe.py).
        shared=True / "all" (faces of EnvCooperationSimplify only, P = 6, and `face` MUST be env's variant-3 face of its
        current states -- the rows are keyed from env's state): the SHARED-ROWS form (csrc/ddz_qnet.h sections 5-6) -- no dense
        GEMM: one row per distinct (rank, face column) of the batch (3.6 % of the 15 T columns at 65,536 tables), first layer +
        ONE k_fc1 rows product over those (K = 288: the table term rides in it), H0[t] = base + the fifteen rows of table t.
        True: the needed rows D stay per table (their first layer skips the ranks no legal move touches); "all" (PolicyLoop's
        default): D as well once per distinct (shared row, count) -- the returned row_index is then the remapped one -- and the
        D chain runs on a side stream beside the H0 chain (self.two_streams; fork / join by events, no host synchronisation).
        Exact per call from the current weights and states (nothing is cached between calls); same values up to fp32 summation
        order in H0 (tests: 1e-5; D and q bit for bit between True and "all").
        The result aliases this object's workspace: consume it before the next call."""
        from . import engine as E
        if self._ver != self._versions():
            self.refresh()
        T, P, H, H1 = face.shape[0], self.P, self.H, self.H1
        if shared and P != 6:
            raise ValueError("shared=True keys the columns of EnvCooperationSimplify's six planes (face variant 3) only")
        if tuple(face.shape[1:]) != (P, 15, 4) or T != env.T or not face.is_cuda:
            raise ValueError(f"face must be a device tensor [T,{P},15,4] of the environment's tables")
        key = ("needed", face.device, T)
        if key not in self._ws:
            FC_TILE = fc_tile()
            cap = (20 * T + 15 * FC_TILE + FC_TILE - 1) // FC_TILE * FC_TILE   # a move takes at most what the actor holds: <= 20 cards
            dev = face.device
            self._ws[key] = {"cap": cap, "y0": None,
                             "dy": torch.zeros((cap, H), dtype=torch.float32, device=dev),
                             "d": torch.zeros((cap, H1), dtype=torch.float32, device=dev),
                             "h0": torch.zeros((T, H1), dtype=torch.float32, device=dev),
                             "row_index": torch.full((T, 64), -1, dtype=torch.int32, device=dev),
                             "seg": torch.zeros(40, dtype=torch.int32, device=dev),
                             "row_cnt": torch.zeros(cap, dtype=torch.uint8, device=dev),
                             "scratch": torch.zeros(E.q_need_scratch_bytes(T), dtype=torch.uint8, device=dev)}
        w = self._ws[key]
        early = shared == "all" and self.two_streams and "side" in w     # the need sets on the side stream, beside the shared rows
        if early:
            cur = torch.cuda.current_stream(face.device)
            w["fork0"].record(cur)
            w["side"].wait_event(w["fork0"])
            with torch.cuda.stream(w["side"]):
                env.q_need(w["cap"], w["scratch"], w["row_index"], w["seg"], w["row_cnt"])
        else:
            env.q_need(w["cap"], w["scratch"], w["row_index"], w["seg"], w["row_cnt"])
        if shared:
            if "srows" not in w:
                FC_TILE = fc_tile()
                scap = (min(15 * T, 4134375) + 15 * FC_TILE + FC_TILE - 1) // FC_TILE * FC_TILE   # cannot overflow (ddz_env.h)
                dev = face.device
                w.update({"scap": scap, "sws": torch.zeros(E.q_shared_ws_bytes(), dtype=torch.uint8, device=dev),
                          "srows": torch.full((T, 16), -1, dtype=torch.int32, device=dev),
                          "srep": torch.full((scap,), -1, dtype=torch.int32, device=dev),
                          "sseg": torch.zeros(40, dtype=torch.int32, device=dev),
                          "ys": torch.zeros((scap, H + 32), dtype=torch.float32, device=dev),
                          "g": torch.zeros((scap, H1), dtype=torch.float32, device=dev)})
                w["y0"] = None                                                     # (1 GB at 65,536 tables: not needed in this form)
            env.q_shared_rows(w["sws"], w["scap"], w["srows"], w["srep"], w["sseg"])

            def h0_chain():
                # G[row] = Y[row] x fc1[rank] + column x Mz[rank] (the table term is linear in the face: folded into the rows --
                # the column rides behind Y in the row, Mz[rank] behind fc1[rank] in the operand: one K = 288 product)
                E.q_features_rows(face, self.Wf, self.bias_f, w["srep"], w["sseg"], w["ys"])
                E.q_fc1_rows_k(w["ys"], w["sseg"], self.W2x, w["g"])
                E.q_gather_h0(w["g"], w["srows"], w["h0"], base=self.base)        # H0[t] = base + sum_r G[row(t, r)]

            if shared == "all":      # the needed rows shared as well: one D row per distinct (shared row, count) (section 6)
                if "dws" not in w:
                    dev = face.device
                    w.update({"dws": torch.zeros(E.q_shared_need_ws_bytes(w["scap"]), dtype=torch.uint8, device=dev),
                              "row_index2": torch.full((T, 64), -1, dtype=torch.int32, device=dev),
                              "drep": torch.full((w["cap"],), -1, dtype=torch.int32, device=dev),
                              "dseg": torch.zeros(40, dtype=torch.int32, device=dev),
                              "drow_cnt": torch.zeros(w["cap"], dtype=torch.uint8, device=dev),
                              "side": torch.cuda.Stream(dev), "fork0": torch.cuda.Event(), "fork": torch.cuda.Event(),
                              "join": torch.cuda.Event()})
                # The H0 chain (first layer of the rows -> G -> gather) and the D chain (D rows -> dY -> D) share only their
                # inputs: both GEMMs are a few hundred tiles -- one or two rounds over the 256 CUs, the launch as long as its
                # last round -- and the bookkeeping kernels are latency-bound, so the D chain runs on a SIDE STREAM beside the
                # H0 chain (fork / join by events: no host synchronisation, capturable in a hipGraph).
                cur = torch.cuda.current_stream(face.device)
                two = self.two_streams
                if two:
                    w["fork"].record(cur)
                    w["side"].wait_event(w["fork"])
                with torch.cuda.stream(w["side"] if two else cur):
                    env.q_shared_need(w["row_index"], w["srows"], w["sseg"], w["scap"], w["dws"], w["cap"], w["row_index2"], w["drep"],
                                      w["dseg"], w["drow_cnt"])
                    E.q_features_drows(face, self.Wf, self.bias_f, self.A, w["srep"], w["drep"], w["dseg"], w["dy"])
                    E.q_fc1_rows(w["dy"], w["dseg"], w["drow_cnt"], self.W2, self.Z, w["d"])
                    if two:
                        w["join"].record(w["side"])
                h0_chain()
                if two:
                    cur.wait_event(w["join"])
                return NeededU(w["h0"], w["d"], w["row_index2"], w["dseg"])
            h0_chain()
            E.q_features_needed(face, self.Wf, self.bias_f, self.A, w["row_index"], None, w["dy"])
            E.q_fc1_rows(w["dy"], w["seg"], w["row_cnt"], self.W2, self.Z, w["d"])
            return NeededU(w["h0"], w["d"], w["row_index"], w["seg"])
        if w["y0"] is None:
            w["y0"] = torch.zeros((T, 15 * H), dtype=torch.float32, device=face.device)
        E.q_features_needed(face, self.Wf, self.bias_f, self.A, w["row_index"], w["y0"], w["dy"])
        torch.addmm(self.base, face.view(T, P * 60), self.Mz_f, out=w["h0"])      # the per-table term (K = 60 P: small)
        if gemm == "mfma":
            E.q_fc1_dense(w["y0"], self.Wd, w["h0"])
        elif gemm == "torch":
            w["h0"].addmm_(w["y0"], self.Wd)
        else:
            raise ValueError("gemm must be 'mfma' or 'torch'")
        E.q_fc1_rows(w["dy"], w["seg"], w["row_cnt"], self.W2, self.Z, w["d"])     # D = dY x fc1[rank] + Z[rank][count]
        return NeededU(w["h0"], w["d"], w["row_index"], w["seg"])

    @staticmethod
    def need_sets(rows, offsets, T):
        """bool [T,15,4]: [t, r, c - 1] <=> some move of table t's CSR list takes exactly c cards of rank r (a joker exists
        once: only c = 1)."""
        N = rows.shape[0]
        pos = torch.arange(N, device=rows.device, dtype=offsets.dtype)
        seg = torch.searchsorted(offsets[1:].contiguous(), pos, right=True).clamp_(max=T - 1).long()
        valid = pos < offsets[T]
        cnt = rows[:, :15].long().clamp(0, 4)
        cnt[:, 13:] = cnt[:, 13:].clamp(max=1)
        hit = (cnt[:, :, None] == torch.arange(1, 5, device=rows.device)[None, None, :]) & valid[:, None, None]   # [N,15,4]
        need = torch.zeros((T, 15, 4), dtype=torch.int32, device=rows.device)
        need.index_add_(0, seg, hit.to(torch.int32))
        return need > 0

    @torch.no_grad()
    def needed_torch(self, face, rows, offsets):
        """The same in plain torch from CSR lists (any device): the statement the engine's needed-rows kernels are tested
        against -- same row layout (rank segments from multiples of fc_tile(), inside a segment table-major then count), so
        row_index and seg compare exactly."""
        if self._ver != self._versions():
            self.refresh()
        T, P, H, H1 = face.shape[0], self.P, self.H, self.H1
        dev = face.device
        FC_TILE = fc_tile()
        need = self.need_sets(rows, offsets, T)                         # [T,15,4]
        per_rank = need.permute(1, 0, 2).reshape(15, T * 4)             # rank-major; inside a rank (t, c) order
        n_r = per_rank.sum(1)
        seg = torch.zeros(40, dtype=torch.int32, device=dev)
        row = 0
        starts = []
        for r in range(15):
            starts.append(row)
            seg[r], seg[16 + r] = row, row // FC_TILE
            row += (int(n_r[r]) + FC_TILE - 1) // FC_TILE * FC_TILE
        seg[15], seg[31], seg[32] = row, row // FC_TILE, int(n_r.sum())
        excl = per_rank.long().cumsum(1) - per_rank.long()
        idx = torch.where(per_rank, excl + torch.tensor(starts, device=dev)[:, None], -1).view(15, T, 4).permute(1, 0, 2)
        row_index = torch.full((T, 64), -1, dtype=torch.int32, device=dev)
        row_index[:, :52] = idx[:, :13].reshape(T, 52).to(torch.int32)
        row_index[:, 52], row_index[:, 53] = idx[:, 13, 0].to(torch.int32), idx[:, 14, 0].to(torch.int32)
        Y = self._first_layer_torch(face)                               # [15,5,T,H]
        y0 = Y[:, 0].permute(1, 0, 2).reshape(T, 15 * H).contiguous()
        dy = torch.zeros((max(row, FC_TILE), H), dtype=torch.float32, device=dev)
        d = torch.zeros((max(row, FC_TILE), H1), dtype=torch.float32, device=dev)
        for r in range(15):
            for c in range(1, 5 if r < 13 else 2):
                dst = idx[:, r, c - 1]
                m = dst >= 0
                dy[dst[m]] = Y[r, c][m] - Y[r, 0][m]
            d[starts[r]: starts[r] + int(n_r[r])] = dy[starts[r]: starts[r] + int(n_r[r])] @ self.W2[r]
            for c in range(1, 5 if r < 13 else 2):                      # the action plane's own term rides on the row
                dst = idx[:, r, c - 1]
                d[dst[dst >= 0]] += self.Z[r, c]
        h0 = torch.addmm(self.base, face.reshape(T, P * 60), self.Mz_f) + y0 @ self.Wd
        nu = NeededU(h0, d, row_index, seg)
        nu.y0, nu.dy = y0, dy
        return nu

    @torch.no_grad()
    def q_csr_needed(self, nu, rows, offsets):
        """q of every CSR row from a NeededU (plain torch; the statement ddz_q_slab_needed is tested against)."""
        T = nu.row_index.shape[0]
        N = rows.shape[0]
        pos = torch.arange(N, device=rows.device, dtype=offsets.dtype)
        seg = torch.searchsorted(offsets[1:].contiguous(), pos, right=True).clamp_(max=T - 1).long()
        cnt = rows[:, :15].long().clamp_(0, 4)
        cnt[:, 13:] = cnt[:, 13:].clamp(max=1)
        r = torch.arange(15, device=rows.device)
        col = torch.where(r[None, :] < 13, 4 * r[None, :] + cnt - 1, 52 + (r[None, :] - 13)).clamp(min=0)
        prow = nu.row_index.long()[seg[:, None], col]                   # [N,15]
        use = (cnt > 0) & (prow >= 0)
        dsum = (nu.d[prow.clamp(min=0)] * use[:, :, None]).sum(1)    # (Z[r][cnt] is part of the row)
        h = nu.h0[seg] + dsum
        return F.relu(h) @ self.w2 + self.b2

    def _first_layer_torch(self, face):
        T, P, H = face.shape[0], self.P, self.H
        X = face.permute(2, 0, 1, 3).reshape(15 * T, P * 4)
        S = torch.addmm(self.bias_f, X, self.Wf).view(15 * T, 4, H)
        return torch.stack([(S + self.A[cnt]).amax(dim=1).view(15, T, H) for cnt in range(5)], dim=1)

    @torch.no_grad()
    def q_csr_packed(self, pu, rows, offsets):
        """q_csr over packed rows (plain torch; the statement ddz_q_slab_packed is tested against)."""
        T = pu.row_index.shape[0]
        N = rows.shape[0]
        pos = torch.arange(N, device=rows.device, dtype=offsets.dtype)
        seg = torch.searchsorted(offsets[1:].contiguous(), pos, right=True).clamp_(max=T - 1).long()
        cnt = rows[:, :15].long().clamp_(0, 4)
        cnt[:, 13:] = cnt[:, 13:].clamp(max=1)
        r = torch.arange(15, device=rows.device)
        row0 = torch.tensor(pu.rank_row0[:15], device=rows.device)
        col = torch.where(r < 13, 4 * r, 52 + (r - 13) - 0)[None, :] + torch.where(r[None, :] < 13, cnt - 1, torch.zeros_like(cnt))
        held = pu.row_index.long()[seg[:, None], col.clamp(min=0)]
        zero_row = row0[None, :] + seg[:, None]
        urow = torch.where((cnt > 0) & (held >= 0), held, zero_row)
        h = pu.u[urow].sum(1) + pu.table_term[seg]
        h = h + F.embedding_bag(r[None, :] * 5 + cnt, self.Z.view(-1, self.H1), mode="sum")
        return F.relu(h) @ self.w2 + self.b2

    @torch.no_grad()
    def q_csr(self, U, rows, offsets):
        """The per-row stage with plain torch ops over CSR lists (the statement the engine's ddz_q_slab is tested
        against): rows int8 [N,16] count rows (ddz_legal / ddz_slab_to_csr; rows beyond offsets[T] are padding and get
        some table's value), offsets int32 [T+1] -> q f32 [N].  No host sync: N is the buffer size."""
        T = U.shape[2]
        N = rows.shape[0]
        pos = torch.arange(N, device=rows.device, dtype=offsets.dtype)
        seg = torch.searchsorted(offsets[1:].contiguous(), pos, right=True).clamp_(max=T - 1).long()
        cnt = rows[:, :15].long().clamp_(0, 4)
        cnt[:, 13:] = cnt[:, 13:].clamp(max=1)                            # a joker exists once
        rc = torch.arange(15, device=rows.device)[None, :] * 5 + cnt      # [N,15]: (rank, count)
        h = F.embedding_bag(rc * T + seg[:, None], U.view(-1, self.H1), mode="sum")
        h = h + F.embedding_bag(rc, self.Z.view(-1, self.H1), mode="sum")
        return F.relu(h) @ self.w2 + self.b2

    @torch.no_grad()
    def q_slab(self, env, U, out=None):
        """The per-row stage over the engine's slab lists (ddz_q_slab): q f32 [T, stride], entries beyond counts[t]
        untouched.  Feeds env.policy_step_slab / select_slab."""
        if isinstance(U, NeededU):
            return env.q_slab_needed(U.h0, U.d, U.row_index, self.w2, self.b2, out=out)
        if isinstance(U, PackedU):
            return env.q_slab_packed(U.u, U.row_index, U.rank_row0, U.table_term, self.Z, self.w2, self.b2, out=out)
        return env.q_slab(U, self.Z, self.w2, self.b2, out=out)


class PackedU:
    """FactorisedQ.tables_packed's result: u f32 [>= n_rows, 256] (fc1's pre-activation contribution of the packed (rank,
    count, table) rows), row_index int32 [T,64], rank_row0 (16 python ints), table_term f32 [T,256]."""
    __slots__ = ("u", "row_index", "rank_row0", "table_term")

    def __init__(self, u, row_index, rank_row0, table_term):
        self.u, self.row_index, self.rank_row0, self.table_term = u, row_index, rank_row0, table_term


class NeededU:
    """FactorisedQ.needed's result: h0 f32 [T,256] (fc1's pre-activation of the pass: every count 0), d f32 [rows,256] (what a
    needed (rank, count >= 1) adds to it, the action plane's weights-only term Z[rank][count] included), row_index int32
    [T,64], seg int32 [40] (device: segment starts, rows in use)."""
    __slots__ = ("h0", "d", "row_index", "seg", "y0", "dy")

    def __init__(self, h0, d, row_index, seg):
        self.h0, self.d, self.row_index, self.seg = h0, d, row_index, seg
        self.y0 = self.dy = None


def ragged_q(net, face, rows, offsets):
    """Q(face_t, action) for every legal row of every table (dqn.py:56,67: policy_net(face, actions) for all tables at
    once): face f32 [T,P,15,4], rows int8 [N,16] + offsets int32 [T+1] in CSR order -> q f32 [N].  The factorised
    tables are cached on the network object and rebuilt when its weights change."""
    fq = getattr(net, "_ddz_factorised", None)
    if fq is None:
        fq = net._ddz_factorised = FactorisedQ(net)
    return fq.q_csr(fq.tables(face), rows, offsets)


class PolicyLoop:
    """game.py:95-104 for T tables with a Q-network on every seat, one lock-step iteration per step(), no per-table work on
    the host:
        face -> Q of every legal move of every table (slab layout) -> ddz_policy_step_slab (epsilon-greedy arg-max + apply +
        next lists + next face, ONE launch).
    mode "needed" (default): FactorisedQ.needed -- the rows legal moves use, found on the device; the per-rank rows GEMM on
        the engine's fp32 MFMA kernel (segment sizes stay in device memory), the plain dense GEMM by hipBLASLt (gemm="mfma":
        by the same MFMA kernel); nothing crosses to the host, every launch is graph-capturable;
    mode "packed": round 3's form (15 + cards-in-hand rows per table, fifteen library GEMMs, one 128-byte device -> host copy
        per iteration for their shapes); mode "full": all 69 rows per table, fixed shapes (packed=True / False select these)."""

    def __init__(self, env, net, face_variant=3, epsilon=0.0, auto_reset=True, packed=None, mode=None, gemm="torch", shared=None):
        from .engine import FACE_PLANES
        if FACE_PLANES[face_variant] != net.planes:
            raise ValueError("the network's input planes do not match the face variant")
        if mode is None:
            mode = "needed" if packed is None else ("packed" if packed else "full")
        if mode not in ("needed", "packed", "full"):
            raise ValueError("mode must be 'needed', 'packed' or 'full'")
        self.env, self.fq = env, FactorisedQ(net)
        self.variant, self.epsilon, self.auto_reset = int(face_variant), float(epsilon), bool(auto_reset)
        self.mode, self.gemm = mode, gemm
        # shared rows (FactorisedQ.needed(shared=True)): the default wherever it applies -- the needed form on
        # EnvCooperationSimplify faces (variant 3), whose columns ddz_q_shared_rows keys from the environment's state
        # (True: H0 from shared rows; "all": the needed rows D shared as well -- the default)
        self.shared = ("all" if (mode == "needed" and int(face_variant) == 3) else False) if shared is None else \
            ("all" if shared == "all" else bool(shared))
        if self.shared and (mode != "needed" or int(face_variant) != 3):
            raise ValueError("shared rows need mode 'needed' and face variant 3")
        T = env.T
        self.face = env.observe(self.variant)
        self.packed = mode == "packed"
        self.U = torch.zeros((15, 5, T, self.fq.H1), dtype=torch.float32, device=env.device) if mode == "full" else None
        self.q = torch.zeros((T, env.slab_stride), dtype=torch.float32, device=env.device)
        self.choice = torch.empty(T, dtype=torch.int32, device=env.device)
        if not env._slab_fresh:
            env.legal_slab()

    def describe(self):
        if self.mode == "needed" and self.shared:
            return ("ddz_q_need (the (rank, count) rows the legal moves use) + ddz_q_shared_rows (one row per DISTINCT (rank, face "
                    "column) of the batch, direct-addressed, on the device) -> ddz_q_features_rows (first layer of the shared rows) + "
                    "ddz_q_features_needed (dY of the needed rows) -> ddz_q_fc1_rows twice (G = Y x fc1[rank] over the shared rows, "
                    "D = dY x fc1[rank] over the needed rows: k_fc1, segment tables in device memory) -> H0 = table term + "
                    "ddz_q_gather_h0 (the fifteen shared rows of every table; no dense K = 3840 GEMM) -> ddz_q_slab_needed -> "
                    "ddz_policy_step_slab(greedy, face): every legal action of every table gets its exact Q value each iteration "
                    "from the current weights; nothing is kept between iterations, nothing crosses to the host")
        if self.mode == "needed":
            return ("ddz_q_need (the (rank, count) rows the legal moves use, on the device) -> ddz_q_features_needed (first "
                    "layer: y0 per table + dY per needed row) -> H0 = tab + y0 x Wd (K = 3840: "
                    + ("torch.addmm / hipBLASLt" if self.gemm == "torch" else "ddz_q_fc1_dense, the fp32 MFMA kernel k_fc1")
                    + ") + ddz_q_fc1_rows (D = dY x fc1[rank], k_fc1 with the segment table in device memory)"
                    + " -> ddz_q_slab_needed -> ddz_policy_step_slab(greedy, face): every legal action of every table gets its Q "
                    "value each iteration; nothing crosses to the host")
        if self.mode == "packed":
            return ("FactorisedQ.tables_packed [ddz_q_features_packed + one torch GEMM per rank over 15 + cards-in-hand rows per "
                    "table] -> ddz_q_slab_packed -> ddz_policy_step_slab; one 128-byte device -> host copy per iteration")
        return "FactorisedQ.tables (all 69 (rank, count) rows per table, fixed shapes) -> ddz_q_slab -> ddz_policy_step_slab"

    def q_values(self):
        """q [T, stride] of the current lists (valid in [:, :counts[t]])"""
        if self.mode == "needed":
            return self.fq.q_slab(self.env, self.fq.needed(self.env, self.face, gemm=self.gemm, shared=self.shared), out=self.q)
        if self.mode == "packed":
            return self.fq.q_slab(self.env, self.fq.tables_packed(self.face, self.env.actor_hands()), out=self.q)
        self.fq.tables(self.face, out=self.U)
        return self.fq.q_slab(self.env, self.U, out=self.q)

    def step(self, traj=None):
        q = self.q_values()
        done, r, illegal, _ = self.env.policy_step_slab(q, self.epsilon, face_variant=self.variant, face_out=self.face,
                                                        choice_out=self.choice, auto_reset=self.auto_reset, traj=traj)
        return done, r, illegal

    def run(self, n):
        for _ in range(int(n)):
            self.step()

    def capture(self, n=1):
        """n lock-step iterations as ONE hipGraph (the needed forms have no host synchronisation and no size-dependent shape):
        returns the torch.cuda.CUDAGraph; every .replay() runs the n iterations on the state the previous ones left (states,
        faces, choices, q values: bit for bit what n eager step() calls give -- tests/test_gpu_qnet.py).  Call step() a few
        times first (workspaces allocated, libraries warm).  An iteration is ~30 short launches: the replay removes the host's
        share of the gaps between them."""
        if self.mode != "needed":
            raise ValueError("only the needed forms are free of host synchronisation")
        dev = self.env.device
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                self.run(n)
        torch.cuda.current_stream(dev).wait_stream(s)
        return g

    def profile(self, n=10):
        """Per-stage device time of n iterations of the needed form (HIP events on the launching stream around every stage)
        with each stage's algorithmic FLOP or bytes: {stage: {"us", "kernel", "flop" | "bytes", "note"}}.  Synchronises."""
        from . import engine as E
        if self.mode != "needed":
            raise ValueError("profile() describes the needed form")
        env, fq, T, P = self.env, self.fq, self.env.T, self.fq.P
        w = None
        names = ("need", "shared_rows", "features_shared", "features", "table_term", "fc1_dense", "fc1_shared", "gather_h0", "shared_need",
                 "fc1_rows", "row_stage", "env_step")
        ev = {k: [] for k in names}
        rows_needed = rows_padded = moves = rows_shared = rows_shared_padded = rows_private = 0

        def timed(name, fn):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            ev[name].append((a, b))

        self.q_values()                                                   # (workspace exists)
        w = fq._ws[("needed", self.face.device, T)]
        for _ in range(int(n)):
            timed("need", lambda: env.q_need(w["cap"], w["scratch"], w["row_index"], w["seg"], w["row_cnt"]))
            if self.shared:
                timed("shared_rows", lambda: env.q_shared_rows(w["sws"], w["scap"], w["srows"], w["srep"], w["sseg"]))
                timed("features_shared", lambda: E.q_features_rows(self.face, fq.Wf, fq.bias_f, w["srep"], w["sseg"], w["ys"]))
                timed("fc1_shared", lambda: E.q_fc1_rows_k(w["ys"], w["sseg"], fq.W2x, w["g"]))
                timed("gather_h0", lambda: E.q_gather_h0(w["g"], w["srows"], w["h0"], base=fq.base))
                if self.shared == "all":
                    timed("shared_need", lambda: env.q_shared_need(w["row_index"], w["srows"], w["sseg"], w["scap"], w["dws"], w["cap"],
                                                                   w["row_index2"], w["drep"], w["dseg"], w["drow_cnt"]))
                    timed("features", lambda: E.q_features_drows(self.face, fq.Wf, fq.bias_f, fq.A, w["srep"], w["drep"], w["dseg"], w["dy"]))
                else:
                    timed("features", lambda: E.q_features_needed(self.face, fq.Wf, fq.bias_f, fq.A, w["row_index"], None, w["dy"]))
                sseg = w["sseg"].cpu()
                rows_shared += int(sseg[32]); rows_shared_padded += int(sseg[15])
            else:
                timed("features", lambda: E.q_features_needed(self.face, fq.Wf, fq.bias_f, fq.A, w["row_index"], w["y0"], w["dy"]))
                timed("table_term", lambda: torch.addmm(fq.base, self.face.view(T, P * 60), fq.Mz_f, out=w["h0"]))
                if self.gemm == "mfma":
                    timed("fc1_dense", lambda: E.q_fc1_dense(w["y0"], fq.Wd, w["h0"]))
                else:
                    timed("fc1_dense", lambda: w["h0"].addmm_(w["y0"], fq.Wd))
            all_ = self.shared == "all"
            sg, rc, ri = (w["dseg"], w["drow_cnt"], w["row_index2"]) if all_ else (w["seg"], w["row_cnt"], w["row_index"])
            timed("fc1_rows", lambda: E.q_fc1_rows(w["dy"], sg, rc, fq.W2, fq.Z, w["d"]))
            timed("row_stage", lambda: env.q_slab_needed(w["h0"], w["d"], ri, fq.w2, fq.b2, out=self.q))
            seg = sg.cpu()
            if all_:
                rows_private += int(w["seg"].cpu()[32])
            rows_needed += int(seg[32]); rows_padded += int(seg[15]); moves += int(env.counts.sum())
            timed("env_step", lambda: env.policy_step_slab(self.q, self.epsilon, face_variant=self.variant, face_out=self.face,
                                                           choice_out=self.choice, auto_reset=self.auto_reset))
        torch.cuda.synchronize(env.device)
        us = {k: sum(a.elapsed_time(b) for a, b in v) * 1e3 / len(v) for k, v in ev.items() if v}
        rn, rp, mv = rows_needed / n, rows_padded / n, moves / n
        H = fq.H
        if self.shared:
            rs, rsp = rows_shared / n, rows_shared_padded / n
            return {
                "need": {"us": us["need"], "kernel": "k_q_need_mask + k_q_need_scan + k_q_need_assign", "bytes": mv * 16 + T * (8 + 8 + 256),
                         "note": "list rows read, need sets written and read, row_index written"},
                "shared_rows": {"us": us["shared_rows"], "kernel": "memset + k_qs_mark + k_qs_count + k_qs_seg + k_qs_assign + k_qs_rows",
                                "bytes": T * 176 + 3 * 4134375 * 4 + T * 16 * 4 * 3 + rs * 8,
                                "note": f"one row per distinct (rank, face column): {rs:.0f} of the {15 * T} columns ({rs / (15 * T):.3f}); "
                                        "state read, the 16.5-MB slot table cleared / counted / assigned, rows [T,16] written"},
                "features_shared": {"us": us["features_shared"], "kernel": "k_q_feat_rows<6>", "bytes": rs * (P * 16 + (H + 32) * 4),
                                    "note": "first layer (count 0) of the shared rows + the table term of their columns (linear in the "
                                            "face: folded into the rows -- no [T, 360] x [360, 256] GEMM per iteration)"},
                "fc1_shared": {"us": us["fc1_shared"], "kernel": "k_fc1<true>", "flop": 2.0 * rs * (H + 32) * H,
                               "note": f"G = [Y | column] x [fc1[rank] ; Mz[rank]] (K = 288) over the {rs:.0f} shared rows ({rsp:.0f} with the padding of the fifteen "
                                       f"segments) -- the dense form of the same term is 2 x {T} x 3840 x 256 = {2.0 * T * 15 * H * H / 1e9:.0f} GFLOP"},
                "gather_h0": {"us": us["gather_h0"], "kernel": "k_qs_gather", "bytes": T * (64 + 2 * H * 4) + rs * H * 4,
                              "note": f"H0 read and written, rows [T,16] read, every row of G once ({rs * H * 4 / 1e6:.0f} MB: the fifteen "
                                      f"1-KB reads per table -- {T * 15 * H * 4 / 1e9:.2f} GB -- are served by L2 / MALL)"},
                **({"shared_need": {"us": us["shared_need"], "kernel": "memset + k_qd_mark + k_qd_count + k_qd_seg + k_qd_assign + k_qd_remap",
                                    "bytes": T * 64 * 4 * 3 + T * 64 + rs * 16 * 3 + rn * 5, "needed_triples": rows_private / n,
                                    "note": f"one D row per distinct (shared row, count) some table needs: {rn:.0f} rows for the "
                                            f"{rows_private / n:.0f} needed (table, rank, count) triples ({rows_private / n / T:.2f} per table)"},
                    "features": {"us": us["features"], "kernel": "k_q_feat_drows<6>", "bytes": rn * (P * 16 + H * 4 + 8),
                                 "note": "dY of the shared D rows"}} if self.shared == "all" else
                   {"features": {"us": us["features"], "kernel": f"k_q_feat_needed<{P}> (y0 = null)", "bytes": T * P * 240 + rn * H * 4 + T * 256,
                                 "note": "face + row_index read, dY [needed rows, 256] written; ranks no legal move touches are skipped"}}),
                "fc1_rows": {"us": us["fc1_rows"], "kernel": "k_fc1<true>", "flop": 2.0 * rn * H * H,
                             "note": f"D = dY x fc1[rank]: {rn:.0f} rows per iteration ({rn / T:.2f} per table), {rp:.0f} computed "
                                     "with the padding of the fifteen tile-aligned segments; FLOP of the rows"},
                "row_stage": {"us": us["row_stage"], "kernel": "k_q_slab_needed", "bytes": T * H * 4 + rn * H * 4 + mv * 20,
                              "note": "H0 + the D rows + the list rows read, q written"},
                "env_step": {"us": us["env_step"], "kernel": "k_slab<4,true>", "bytes": T * (2 * 176 + P * 240 + 8) + mv * 24,
                             "note": "arg-max over q, apply, new lists, new face"},
            }
        return {
            "need": {"us": us["need"], "kernel": "k_q_need_mask + k_q_need_scan + k_q_need_assign", "bytes": mv * 16 + T * (8 + 8 + 256),
                     "note": "list rows read, need sets written and read, row_index written"},
            "features": {"us": us["features"], "kernel": f"k_q_feat_needed<{P}>", "bytes": T * P * 240 + T * 15 * H * 4 + rn * H * 4 + T * 256,
                         "note": "face + row_index read, y0 [T, 3840] + dY [needed rows, 256] written"},
            "table_term": {"us": us["table_term"], "kernel": "torch.addmm (hipBLASLt)", "flop": 2.0 * T * P * 60 * H,
                           "note": "fc1 bias + the face part of conv_shunzi: [T, 60 P] x [60 P, 256]"},
            "fc1_dense": {"us": us["fc1_dense"], "kernel": "k_fc1<false>" if self.gemm == "mfma" else "torch.addmm (hipBLASLt)",
                          "flop": 2.0 * T * 15 * H * H, "note": "H0 += y0 [T, 3840] x Wd [3840, 256]"},
            "fc1_rows": {"us": us["fc1_rows"], "kernel": "k_fc1<true>", "flop": 2.0 * rn * H * H,
                         "note": f"D = dY x fc1[rank]: {rn:.0f} needed rows per iteration ({rn / T:.2f} per table), {rp:.0f} computed "
                                 "with the padding of the fifteen tile-aligned (256-row) segments; FLOP of the needed rows"},
            "row_stage": {"us": us["row_stage"], "kernel": "k_q_slab_needed", "bytes": T * H * 4 + rn * H * 4 + mv * 20,
                          "note": "H0 + the needed D rows + the list rows read, q written"},
            "env_step": {"us": us["env_step"], "kernel": "k_slab<4,true>", "bytes": T * (2 * 176 + P * 240 + 8) + mv * 24,
                         "note": "arg-max over q, apply, new lists, new face"},
        }

    def variants(self, timed_loop, sync):
        """env steps/s of the other forms of the same loop on the same environment (bench.py): the dense GEMM by hipBLASLt,
        round 3's packed rows, fixed shapes."""
        out = {}
        T = self.env.T
        other = "mfma" if self.gemm == "torch" else "torch"
        forms = [("needed_dense_gemm_by_" + ("k_fc1" if other == "mfma" else "hipblaslt"), {"mode": "needed", "gemm": other, "shared": False})]
        if self.shared:   # the dense form of H0 (round 4's first form: one K = 3840 GEMM over every table)
            forms.insert(0, ("needed_dense_gemm_by_" + ("hipblaslt" if self.gemm == "torch" else "k_fc1"),
                             {"mode": "needed", "gemm": self.gemm, "shared": False}))
        for name, kw in (*forms,
                         ("packed_rows_round3", {"mode": "packed"}), ("fixed_shapes", {"mode": "full"})):
            loop = PolicyLoop(self.env, self.fq.net, face_variant=self.variant, epsilon=self.epsilon, **kw)
            loop.run(2)
            dt, reps = timed_loop(lambda: loop.run(5), sync, min_s=0.2, max_reps=64)
            out[name + "_env_steps_per_s"] = T * 5 * reps / dt
            del loop
        return out


class Replay:
    """Ring buffer of transitions on the device (dqn.py:11,22-23: deque(maxlen=REPLAY_SIZE) of tuples)."""

    def __init__(self, size, planes, device):
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)  # noqa: E731
        self.s0, self.a0, self.s1, self.a1 = z(size, planes, 15, 4), z(size, 15, 4), z(size, planes, 15, 4), z(size, 15, 4)
        self.r, self.done = z(size), torch.zeros(size, dtype=torch.bool, device=device)
        self.size, self.n, self.head = size, 0, 0

    def push(self, tr):
        k = tr["reward"].numel()
        if k == 0:
            return
        if k > self.size:
            tr = {key: v[-self.size:] for key, v in tr.items()}
            k = self.size
        idx = (self.head + torch.arange(k, device=self.r.device)) % self.size
        self.s0[idx], self.a0[idx], self.s1[idx], self.a1[idx] = tr["s0"], tr["a0"], tr["s1"], tr["a1"]
        self.r[idx], self.done[idx] = tr["reward"], tr["done"]
        self.head = (self.head + k) % self.size
        self.n = min(self.size, self.n + k)

    def sample(self, k):
        idx = torch.randint(0, self.n, (k,), device=self.r.device)
        return {"s0": self.s0[idx], "a0": self.a0[idx], "s1": self.s1[idx], "a1": self.a1[idx],
                "reward": self.r[idx], "done": self.done[idx]}


EPSILON_HIGH, EPSILON_LOW, DECAY = 0.5, 0.01, int((8000 * (2 / 3)) / 5)   # config.py:9-13


def epsilon_schedule(episode, high=EPSILON_HIGH, low=EPSILON_LOW, decay=DECAY):
    """DQNFirst.update_epsilon (dqn.py:73-76): low + (high - low) * exp(-episode / decay)."""
    import math
    return low + (high - low) * math.exp(-1.0 * episode / decay)


def td_step(policy, target, optimizer, batch, gamma=0.95):
    """One perceive() update (dqn.py:33-48): y = r + (1 - done) * gamma * Q_target(s1, a1), MSE against
    Q_policy(s0, a0), one optimizer step.  Returns the loss (a tensor: no host sync)."""
    with torch.no_grad():
        y = td_target(batch, target(batch["s1"], batch["a1"]), gamma)
    loss = F.mse_loss(policy(batch["s0"], batch["a0"]).view(-1), y)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    return loss.detach()
