"""BatchedEnv: T independent Doudizhu tables advanced in lock-step on one MI355X.

Host-side mirror of the reference's env adapter (envi.py:16-161) re-cut as batched verbs.
PyTorch owns every buffer (state, scratch, CSR list, outputs); the HIP library
(csrc/, C ABI include/ddz_env.h) is handed raw device pointers and the current stream.
Nothing here computes game logic on the CPU, and nothing falls back to it.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import DdzError, check

ROW = 16
NFIELDS = 11
TRAJ_BYTES = 32
NUM_ACTIONS = 13527
STEP_RANDOM, STEP_CHOICE, STEP_ROWS, STEP_IDS = 0, 1, 2, 3
FACE_PLANES = (4, 7, 9, 6)  # Env, EnvComplicated, EnvCooperation, EnvCooperationSimplify
F_HAND0, F_HIST0, F_RECENT0, F_TAKEN, F_META = 0, 3, 6, 9, 10
# a 20-card hand never has more than this many legal moves (tests/test_rules_bounds.py);
# the default row capacity is T * MAX_LEGAL_PER_TABLE so the CSR list cannot overflow.
MAX_LEGAL_PER_TABLE = 512
CSR_STAGING_BYTES = 16 << 30  # default staging budget of rollout_random_csr (slabs of a batch of iterations)


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise DdzError("no MI355X visible (torch.cuda.is_available() is False): the engine has "
                       "no CPU fallback")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise DdzError(f"device must be a cuda (ROCm) device, got {dev}")
    return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    """the caller's current HIP stream on `device` as a raw pointer (every launch of the library goes there)"""
    if _raw_stream is not None and device.index is not None:
        return C.c_void_p(_raw_stream(device.index))      # (~0.3 us; torch.cuda.current_stream() builds a Stream object: ~4 us)
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class BatchedEnv:
    """T tables; table t is global table `table_id_base + t` (keys the RNG)."""

    def __init__(self, n_tables, seed=0, device="cuda:0", table_id_base=0, row_capacity=None,
                 want_ids=True, native_joker_kickers=False, host_mirror=False, _debug_tables_per_wave=None,
                 _debug_slab_coop=None, _debug_slab_work_list=None, _debug_auto_teams=None):
        # native_joker_kickers: the optional rule set with the 24 extra rows the reference's native
        # get_moves is known to emit (server/mcts/get_moves.py:22-34); default off = exactly card.py
        self.native_joker_kickers = bool(native_joker_kickers)
        self.lib = _lib.lib(jk=self.native_joker_kickers)
        self.device = _require_gpu(device)
        self.T = int(n_tables)
        if self.T <= 0:
            raise ValueError("n_tables must be positive")
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.table_id_base = int(table_id_base)
        self.cap = int(row_capacity) if row_capacity is not None else self.T * MAX_LEGAL_PER_TABLE
        if self.cap > 0x7FFFFFFF:
            raise ValueError("row capacity must be indexable with int32")
        d = self.device
        # host_mirror (the N = 1 `Env` view): the state rows and the list sizes live in PINNED HOST memory -- the kernels
        # read and write them over the host link (a few hundred bytes per launch), the host reads them after one stream
        # synchronisation with no copy at all.  Latency trade for one table; never for a batch.
        self.host_mirror = bool(host_mirror)
        if self.host_mirror:
            self.state = torch.zeros(self.lib.ddz_state_bytes(self.T), dtype=torch.uint8).pin_memory()
        else:
            self.state = torch.zeros(self.lib.ddz_state_bytes(self.T), dtype=torch.uint8, device=d)
        self.scratch = torch.zeros(self.lib.ddz_scratch_bytes(self.T), dtype=torch.uint8, device=d)
        self.offsets = torch.zeros(self.T + 1, dtype=torch.int32, device=d)
        self.rows = torch.zeros((self.cap, ROW), dtype=torch.int8, device=d)
        self.ids = torch.zeros(self.cap, dtype=torch.int32, device=d) if want_ids else None
        if self.host_mirror:
            self.counts = torch.zeros(self.T, dtype=torch.int32).pin_memory()
        else:
            self.counts = torch.zeros(self.T, dtype=torch.int32, device=d)  # slab layout: list sizes
        self.done = torch.zeros(self.T, dtype=torch.uint8, device=d)
        self.reward = torch.zeros(self.T, dtype=torch.int8, device=d)
        self.illegal = torch.zeros(self.T, dtype=torch.uint8, device=d)
        self._stats = torch.zeros(8, dtype=torch.int64, device=d)
        self._legal_fresh = self._slab_fresh = self._csr_fresh = False
        # raw pointers of the persistent buffers, converted once (the host side of a call is ~10 us of ctypes work)
        self._pp = {k: _p(getattr(self, k)) for k in ("counts", "rows", "ids", "done", "reward", "illegal", "offsets")}
        h = C.c_void_p()
        check(self.lib.ddz_create(C.byref(h), self.T, self.seed, self.table_id_base, d.index,
                                  _p(self.state), self.state.numel(), _p(self.scratch),
                                  self.scratch.numel()))
        self._h = h
        if _debug_tables_per_wave is not None or _debug_slab_coop is not None or _debug_slab_work_list is not None:
            # test hook: results never depend on the launch geometry
            check(self.lib.ddz_debug_set_geometry(h, int(_debug_tables_per_wave or 0),
                                                  -1 if _debug_slab_coop is None else int(_debug_slab_coop),
                                                  -1 if _debug_slab_work_list is None else int(bool(_debug_slab_work_list))))
        if _debug_auto_teams is not None:
            # test hook: auto_choose's wavefronts without tables help their workgroup's running searches (default) or not
            check(self.lib.ddz_debug_set_auto_teams(h, int(_debug_auto_teams)))   # 0 off, 1 default, 2 teams without team-first

    def close(self):
        if getattr(self, "_h", None):
            self.lib.ddz_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- views of the packed state ([T,16] int8/uint8, zero-copy) ----
    def field(self, f):
        return self.state.view(self.T, NFIELDS, ROW)[:, f]

    @property
    def role(self):
        return self.field(F_META)[:, 0]

    def hands(self):
        return self.state.view(self.T, NFIELDS, ROW)[:, F_HAND0:F_HAND0 + 3]

    # ---- verbs ----
    def reset(self, mask=None):
        """Env.reset() + prepare() (envi.py:30-36, game.py:170-171) for masked tables."""
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if mask.numel() != self.T:
                raise ValueError("mask must have one byte per table")
        check(self.lib.ddz_reset(self._h, _p(mask), _stream(self.device)))
        self._legal_fresh = self._slab_fresh = self._csr_fresh = False

    def legal(self):
        """CSR legal-move lists of all tables (envi.py:98-116 valid_actions(tensor=False)).
        Returns (offsets[T+1] i32, rows[cap,16] i8, ids[cap] i32 | None); only the first
        offsets[T] rows are meaningful.  No host sync."""
        check(self.lib.ddz_legal(self._h, _p(self.offsets), _p(self.rows), _p(self.ids), self.cap,
                                 _stream(self.device)))
        self._legal_fresh, self._slab_fresh, self._csr_fresh = True, False, False  # legal() packed CSR rows into the row buffer
        return self.offsets, self.rows, self.ids

    def _need_legal(self):
        if not self._legal_fresh:
            self.legal()

    def step(self, sel=None, mode=STEP_CHOICE, auto_reset=True, traj=None):
        """Apply one action per table (envi.py:63-70 step_manual / :79-85 step_random).
        mode STEP_RANDOM: sel ignored; STEP_CHOICE: sel int32[T] index into each legal
        segment; STEP_ROWS: sel int8[T,16] count rows; STEP_IDS: sel int32[T] canonical action ids
        (-1 = engine RNG for that table), what auto_choose() returns.  Returns (done u8[T], r i8[T],
        illegal u8[T]); r = -1 lord won, +1 farmers won (rule_play.py:14)."""
        self._need_legal()
        if mode in (STEP_CHOICE, STEP_IDS):
            sel = sel.to(device=self.device, dtype=torch.int32).contiguous()
            if sel.numel() != self.T:
                raise ValueError("choice / ids must have one entry per table")
        elif mode == STEP_ROWS:
            sel = sel.to(device=self.device, dtype=torch.int8).contiguous()
            if tuple(sel.shape) != (self.T, ROW):
                raise ValueError("rows must be [T,16] int8")
        elif mode != STEP_RANDOM:
            raise ValueError("bad step mode")
        if traj is not None and (traj.dtype != torch.uint8 or traj.numel() != self.T * TRAJ_BYTES
                                 or not traj.is_contiguous()):
            raise ValueError("traj must be a contiguous uint8 [T,32] tensor")
        check(self.lib.ddz_step(self._h, mode, _p(sel) if mode != STEP_RANDOM else None,
                                _p(self.offsets), _p(self.rows), int(bool(auto_reset)),
                                _p(self.done), _p(self.reward), _p(self.illegal), _p(traj),
                                _stream(self.device)))
        self._legal_fresh = self._slab_fresh = self._csr_fresh = False
        return self.done, self.reward, self.illegal

    def step_onehot(self, actions, auto_reset=True, traj=None):
        """step_manual with the reference's action encoding (envi.py:63-70): actions f32/int
        [T,15,4] thermometers (what valid_actions(tensor=True) rows look like); decoded on the
        device by the row sum of onehot2arr (envi.py:148-157)."""
        a = actions.to(self.device)
        if tuple(a.shape) != (self.T, 15, 4):
            raise ValueError("actions must be [T,15,4]")
        rows = torch.zeros((self.T, ROW), dtype=torch.int8, device=self.device)
        rows[:, :15] = a.sum(dim=2).round().to(torch.int8)
        return self.step(rows, STEP_ROWS, auto_reset, traj)

    def step_random(self, auto_reset=True, traj=None):
        return self.step(None, STEP_RANDOM, auto_reset, traj)

    # ---- rule-based opponent (envi.py:72-77 step_auto) ----
    def auto_choose(self, auto_roles=0b101, out=None, stats=None, _debug_kernel=None):
        """The rule agent's move (RuleBasedModel.choose, rule_based/utils/rule_based_model.py:43-101) for every table
        whose actor's role bit is set in auto_roles (bit 0 up, 1 lord, 2 down; default: both farmers): int32[T]
        canonical action ids, -1 for the other tables (-2 = DDZ_AUTO_INVALID: a state the agent cannot decide, which
        DDZ_STEP_IDS flags illegal -- never a silent random move).  stats: optional int64 [T,2] {combinations, search nodes} of the full
        enumeration; without it the kernel runs an exact branch and bound (same ids, ~3x faster)."""
        if out is None:
            out = torch.empty(self.T, dtype=torch.int32, device=self.device)
        if stats is not None and (stats.dtype != torch.int64 or stats.numel() != 2 * self.T or not stats.is_contiguous()):
            raise ValueError("stats must be a contiguous int64 [T,2] tensor")
        if _debug_kernel is not None:  # test hook: 1 = the sequential cross-check kernel, 2 = the product kernel
            check(self.lib.ddz_debug_auto_choose_state(self._h, int(_debug_kernel), int(auto_roles), _p(out), _p(stats),
                                                       _stream(self.device)))
        else:
            check(self.lib.ddz_auto_choose_state(self._h, int(auto_roles), _p(out), _p(stats), _stream(self.device)))
        return out

    def step_auto(self, auto_roles=0b101, ids=None, auto_reset=True, traj=None, slab=False):
        """One lock-step iteration in which the roles of auto_roles are played by the rule agent (Env.step_auto,
        envi.py:72-77) and every other table moves by `ids` (int32[T] canonical action ids of a policy; None or -1 =
        engine RNG, i.e. step_random).  game.py:106: `_, done, _ = self.env.step_auto()` for a role without a network."""
        sel = self.auto_choose(auto_roles)
        if ids is not None:
            sel = torch.where(sel >= 0, sel, ids.to(device=self.device, dtype=torch.int32))
        return (self.step_slab if slab else self.step)(sel, STEP_IDS, auto_reset, traj)

    def observe(self, variant=3, out=None):
        """`face` of every table: f32 [T,P,15,4] (envi.py:87-96,165-217)."""
        P = FACE_PLANES[variant]
        if out is None:
            out = torch.empty((self.T, P, 15, 4), dtype=torch.float32, device=self.device)
        elif out.dtype != torch.float32 or out.numel() != self.T * P * 60 or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous float32 [T,{P},15,4] tensor")
        check(self.lib.ddz_observe(self._h, int(variant), _p(out), _stream(self.device)))
        return out

    def select(self, q, epsilon=0.0, out=None):
        """greedy / epsilon-greedy choice per table from per-row values q (f32, CSR order of
        the current legal list; dqn.py:50-71).  Returns int32[T] indices for step(STEP_CHOICE)."""
        if not self._csr_fresh:   # after slab_to_csr() the offsets already describe the current lists: no ddz_legal
            self._need_legal()
        q = q.to(device=self.device, dtype=torch.float32).contiguous().view(-1)
        if out is None:
            out = torch.empty(self.T, dtype=torch.int32, device=self.device)
        check(self.lib.ddz_select(self._h, _p(q), _p(self.offsets), float(epsilon), _p(out),
                                  _stream(self.device)))
        return out

    def select_slab(self, q, epsilon=0.0, out=None):
        """select() for the slab layout: q f32 [T, stride] (entries beyond counts[t] are ignored); returns
        int32[T] indices for step_slab(STEP_CHOICE).  Same greedy / epsilon-greedy rule and RNG as select()."""
        if not self._slab_fresh:
            self.legal_slab()
        if q.dtype != torch.float32 or q.device != self.device or not q.is_contiguous():
            q = q.to(device=self.device, dtype=torch.float32).contiguous()
        if q.numel() != self.T * self.slab_stride:
            raise ValueError("q must be [T, stride]")
        if out is None:
            out = torch.empty(self.T, dtype=torch.int32, device=self.device)
        check(self.lib.ddz_select_slab(self._h, _p(q), self._pp["counts"], self.slab_stride, float(epsilon), _p(out),
                                       _stream(self.device)))
        return out

    def q_slab(self, u, z, w2, b2, out=None):
        """Per-row stage of the ragged Q forward over the current slab lists (ddz_q_slab; dqn_glue.FactorisedQ):
        u f32 [15,5,T,256], z f32 [15,5,256], w2 f32 [256], b2 f32 [1] (device tensors) -> q f32 [T, stride], valid in
        [:, :counts[t]]."""
        if not self._slab_fresh:
            self.legal_slab()
        H = int(u.shape[-1])
        if u.dtype != torch.float32 or tuple(u.shape) != (15, 5, self.T, H) or not u.is_contiguous() or u.device != self.device:
            raise ValueError("u must be a contiguous float32 [15,5,T,hidden] tensor on the engine's device")
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        w2 = w2.to(device=self.device, dtype=torch.float32).contiguous().view(-1)
        b2 = b2.to(device=self.device, dtype=torch.float32).contiguous().view(-1)
        if w2.numel() != H or b2.numel() != 1 or z.numel() != 75 * H:
            raise ValueError("z must be [15,5,hidden], w2 [hidden], b2 [1]")
        if out is None:
            out = torch.zeros((self.T, self.slab_stride), dtype=torch.float32, device=self.device)
        elif out.dtype != torch.float32 or out.numel() != self.T * self.slab_stride or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 [T, stride] tensor")
        check(self.lib.ddz_q_slab(self._h, _p(u), _p(z), H, _p(w2), _p(b2), self._pp["counts"], self._pp["rows"],
                                  self.slab_stride, _p(out), _stream(self.device)))
        return out

    def actor_hands(self):
        """int64 [T,15]: the cards the acting role of every table holds (the state's hand row of the role in the meta
        row) -- what a legal move can take at most of each rank."""
        s = self.state.view(self.T, 11, 16)
        role = s[:, F_META, 0].long().clamp_(max=2)
        return s[:, F_HAND0:F_HAND0 + 3, :15].gather(1, role[:, None, None].expand(self.T, 1, 15))[:, 0].long()

    def q_slab_packed(self, u, row_index, rank_row0, table_term, z, w2, b2, out=None):
        """ddz_q_slab_packed: the per-row stage over packed rows (dqn_glue.FactorisedQ.pack / tables_packed):
        u f32 [n_rows,256], row_index int32 [T,64], rank_row0 = 16 host ints (the ranks' first rows + n_rows),
        table_term f32 [T,256] | None -> q f32 [T, stride], valid in [:, :counts[t]]."""
        if not self._slab_fresh:
            self.legal_slab()
        H = int(u.shape[-1])
        n_rows = int(rank_row0[15])
        if u.dtype != torch.float32 or u.dim() != 2 or u.shape[0] < n_rows or not u.is_contiguous() or u.device != self.device:
            raise ValueError("u must be a contiguous float32 [>= n_rows, hidden] tensor on the engine's device")
        if (row_index.dtype != torch.int32 or tuple(row_index.shape) != (self.T, 64) or not row_index.is_contiguous()
                or row_index.device != self.device):
            raise ValueError("row_index must be a contiguous int32 [T,64] tensor on the engine's device")
        if table_term is not None and (table_term.dtype != torch.float32 or tuple(table_term.shape) != (self.T, H)
                                       or not table_term.is_contiguous() or table_term.device != self.device):
            raise ValueError("table_term must be a contiguous float32 [T,hidden] tensor on the engine's device")
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        w2 = w2.to(device=self.device, dtype=torch.float32).contiguous().view(-1)
        b2 = b2.to(device=self.device, dtype=torch.float32).contiguous().view(-1)
        if w2.numel() != H or b2.numel() != 1 or z.numel() != 75 * H:
            raise ValueError("z must be [15,5,hidden], w2 [hidden], b2 [1]")
        if out is None:
            out = torch.zeros((self.T, self.slab_stride), dtype=torch.float32, device=self.device)
        elif out.dtype != torch.float32 or out.numel() != self.T * self.slab_stride or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 [T, stride] tensor")
        r0 = (C.c_int64 * 15)(*[int(x) for x in rank_row0[:15]])
        check(self.lib.ddz_q_slab_packed(self._h, _p(u), _p(row_index), r0, n_rows,
                                         _p(table_term) if table_term is not None else None, _p(z), H, _p(w2), _p(b2),
                                         self._pp["counts"], self._pp["rows"], self.slab_stride, _p(out), _stream(self.device)))
        return out

    # ---- the needed-rows form of the ragged Q forward (csrc/ddz_qnet.h; dqn_glue.FactorisedQ.needed) ----
    def q_need(self, row_capacity, scratch, row_index, seg, row_cnt):
        """ddz_q_need: which (rank, count >= 1) rows the CURRENT slab lists use, per table, laid out in rank segments:
        writes row_index int32 [T,64], seg int32 [40] and row_cnt uint8 [row_capacity] (device); nothing crosses to the host."""
        if not self._slab_fresh:
            self.legal_slab()
        if (row_index.dtype != torch.int32 or tuple(row_index.shape) != (self.T, 64) or not row_index.is_contiguous()
                or seg.dtype != torch.int32 or seg.numel() < 40 or scratch.dtype != torch.uint8
                or row_cnt.dtype != torch.uint8 or row_cnt.numel() < int(row_capacity)):
            raise ValueError("row_index must be int32 [T,64], seg int32 [40], scratch uint8, row_cnt uint8 [row_capacity]")
        check(self.lib.ddz_q_need(self._h, self._pp["counts"], self._pp["rows"], self.slab_stride, int(row_capacity),
                                  _p(scratch), scratch.numel(), _p(row_index), _p(seg), _p(row_cnt), _stream(self.device)))

    def q_shared_rows(self, ws, row_capacity, rows, rep, seg):
        """ddz_q_shared_rows: one row per DISTINCT (rank, face column) of the CURRENT states (EnvCooperationSimplify faces):
        rows int32 [T,16] (row of (t, r)), rep int32 [row_capacity] (row -> instance 16 t + r), seg int32 [40] (rank segments);
        ws uint8 [q_shared_ws_bytes()].  Nothing crosses to the host."""
        if (rows.dtype != torch.int32 or tuple(rows.shape) != (self.T, 16) or not rows.is_contiguous() or rep.dtype != torch.int32
                or rep.numel() < int(row_capacity) or seg.dtype != torch.int32 or seg.numel() < 40 or ws.dtype != torch.uint8):
            raise ValueError("rows must be int32 [T,16], rep int32 [row_capacity], seg int32 [40], ws uint8")
        check(self.lib.ddz_q_shared_rows(self._h, _p(ws), ws.numel(), int(row_capacity), _p(rows), _p(rep), _p(seg),
                                         _stream(self.device)))

    def q_shared_need(self, row_index, rows, sseg, shared_row_capacity, ws, row_capacity, row_index2, drep, dseg, row_cnt):
        """ddz_q_shared_need: one D row per distinct (shared row, count) some table needs: row_index2 int32 [T,64], drep int32
        [row_capacity], dseg int32 [40], row_cnt uint8 [row_capacity] from q_need's row_index and q_shared_rows' rows / sseg."""
        for x, shp in ((row_index, (self.T, 64)), (row_index2, (self.T, 64)), (rows, (self.T, 16))):
            if x.dtype != torch.int32 or tuple(x.shape) != shp or not x.is_contiguous():
                raise ValueError("row_index / row_index2 must be int32 [T,64], rows int32 [T,16]")
        if (drep.dtype != torch.int32 or drep.numel() < int(row_capacity) or row_cnt.dtype != torch.uint8 or row_cnt.numel() < int(row_capacity)
                or dseg.dtype != torch.int32 or dseg.numel() < 40 or sseg.dtype != torch.int32 or sseg.numel() < 40 or ws.dtype != torch.uint8):
            raise ValueError("drep int32 [row_capacity], row_cnt uint8 [row_capacity], dseg / sseg int32 [40], ws uint8")
        check(self.lib.ddz_q_shared_need(self._h, _p(row_index), _p(rows), _p(sseg), int(shared_row_capacity), _p(ws), ws.numel(),
                                         int(row_capacity), _p(row_index2), _p(drep), _p(dseg), _p(row_cnt), _stream(self.device)))

    def q_slab_needed(self, h0, d, row_index, w2, b2, out=None):
        """ddz_q_slab_needed: q f32 [T, stride] of every legal move from h0 f32 [T,256], d f32 [rows,256] (with z folded in by
        q_fc1_rows), row_index."""
        if not self._slab_fresh:
            self.legal_slab()
        H = int(h0.shape[-1])
        for x, shp in ((h0, (self.T, H)), (d, (d.shape[0], H))):
            if x.dtype != torch.float32 or tuple(x.shape) != shp or not x.is_contiguous() or x.device != self.device:
                raise ValueError("h0 must be float32 [T,hidden], d float32 [rows,hidden], contiguous, on the engine's device")
        if out is None:
            out = torch.zeros((self.T, self.slab_stride), dtype=torch.float32, device=self.device)
        elif out.dtype != torch.float32 or out.numel() != self.T * self.slab_stride or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 [T, stride] tensor")
        check(self.lib.ddz_q_slab_needed(self._h, _p(h0), _p(d), int(d.shape[0]), _p(row_index), H, _p(w2), _p(b2),
                                         self._pp["counts"], self._pp["rows"], self.slab_stride, _p(out), _stream(self.device)))
        return out

    def legal_onehot(self):
        """valid_actions(tensor=True) for all tables: f32 [sum A,15,4] (one host sync)."""
        self._need_legal()
        total = int(self.offsets[-1].item())
        return rows_to_onehot(self.rows[:total])

    # ---- random-policy rollout, slab layout ----
    @property
    def slab_stride(self):
        """rows per table slab when self.rows is used in the slab layout"""
        return self.cap // self.T

    def slab_rows(self):
        """[T, stride, 16] view of the list buffer after rollout_random: table t's legal
        moves are slab_rows()[t, :counts[t]] (ascending canonical id)."""
        st = self.slab_stride
        return self.rows[: self.T * st].view(self.T, st, ROW)

    def slab_ids(self):
        st = self.slab_stride
        return None if self.ids is None else self.ids[: self.T * st].view(self.T, st)

    def legal_mask(self, out=None, unpack=False):
        """get_mask (rule_based/utils/utils.py:45-63) of every table: the legal moves as a dense mask over the
        action space.  Bit-packed int32 [T, 424] (bit id & 31 of word id >> 5; bit 0 = pass), or with
        unpack=True a bool [T, n_actions] tensor (a policy head with one logit per action masks with it)."""
        W = self.lib.ddz_mask_words()
        if out is None:
            out = torch.empty((self.T, W), dtype=torch.int32, device=self.device)
        elif out.dtype != torch.int32 or tuple(out.shape) != (self.T, W) or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous int32 [T,{W}] tensor")
        check(self.lib.ddz_legal_mask(self._h, _p(out), _stream(self.device)))
        if not unpack:
            return out
        bits = (out.unsqueeze(-1) >> torch.arange(32, device=self.device, dtype=torch.int32)) & 1
        return bits.view(self.T, W * 32)[:, : self.lib.ddz_num_actions()].bool()

    def legal_slab(self):
        """Legal-move lists of all tables in the slab layout: (counts[T] i32, rows [T,stride,16] i8,
        ids [T,stride] i32 | None); table t's moves are rows[t, :counts[t]] in ascending canonical id.
        No cross-table prefix, so step_slab() can apply the choices AND write the next lists in one launch."""
        if self.slab_stride < MAX_LEGAL_PER_TABLE:
            raise ValueError(f"slab lists need row_capacity >= {MAX_LEGAL_PER_TABLE} * n_tables")
        check(self.lib.ddz_legal_slab(self._h, _p(self.counts), _p(self.rows), _p(self.ids), self.slab_stride,
                                      _stream(self.device)))
        self._legal_fresh = self._csr_fresh = False  # the CSR buffers (offsets) do not describe the row buffer any more
        self._slab_fresh = True
        return self.counts, self.slab_rows(), self.slab_ids()

    def sync(self):
        """hipStreamSynchronize on the current stream (ddz_sync): after it a host_mirror environment's state / counts are
        what the last launch left."""
        check(self.lib.ddz_sync(self.device.index, _stream(self.device)))

    def step_slab(self, sel=None, mode=STEP_CHOICE, auto_reset=True, traj=None):
        """One lock-step iteration in ONE launch: apply `sel` (STEP_CHOICE: int32[T] index into each
        table's slab list; STEP_ROWS: int8[T,16]; STEP_RANDOM: engine RNG) to the lists legal_slab() /
        the previous step_slab() left in the buffers, then overwrite them with the lists of the new
        states (game.py:95-106 + envi.py:98-116).  Returns (done, r, illegal) like step()."""
        if not self._slab_fresh:
            self.legal_slab()
        pinned = sel is not None and sel.device.type == "cpu" and sel.is_pinned()   # (pinned host memory is device-readable)
        if mode in (STEP_CHOICE, STEP_IDS):
            if sel.dtype != torch.int32 or (sel.device != self.device and not pinned) or not sel.is_contiguous():  # (the host
                sel = sel.to(device=self.device, dtype=torch.int32).contiguous()               # side of a call costs ~10 us)
            if sel.numel() != self.T:
                raise ValueError("choice / ids must have one entry per table")
        elif mode == STEP_ROWS:
            if sel.dtype != torch.int8 or (sel.device != self.device and not pinned) or not sel.is_contiguous():
                sel = sel.to(device=self.device, dtype=torch.int8).contiguous()
            if tuple(sel.shape) != (self.T, ROW):
                raise ValueError("rows must be [T,16] int8")
        elif mode != STEP_RANDOM:
            raise ValueError("bad step mode")
        if traj is not None and (traj.dtype != torch.uint8 or traj.numel() != self.T * TRAJ_BYTES
                                 or not traj.is_contiguous()):
            raise ValueError("traj must be a contiguous uint8 [T,32] tensor")
        pp = self._pp
        check(self.lib.ddz_step_slab(self._h, mode, _p(sel) if mode != STEP_RANDOM else None, pp["counts"],
                                     pp["rows"], pp["ids"], self.slab_stride, 1 if auto_reset else 0,
                                     pp["done"], pp["reward"], pp["illegal"], _p(traj),
                                     _stream(self.device)))
        self._legal_fresh, self._slab_fresh, self._csr_fresh = False, True, False  # the buffers hold the lists of the new states
        return self.done, self.reward, self.illegal

    def policy_step_slab(self, q, epsilon=0.0, face_variant=None, face_out=None, choice_out=None, auto_reset=True,
                         traj=None):
        """The environment side of one lock-step iteration of a value-based policy in ONE launch: (epsilon-)greedy
        arg-max of q [T, stride] over each table's slab list (= select_slab), apply it (= step_slab), write the new
        lists and, with face_variant, the `face` [T,P,15,4] of the new states (= observe).  Returns (done, r, illegal,
        face | None); bit-identical to the three separate calls."""
        if not self._slab_fresh:
            self.legal_slab()
        if q.dtype != torch.float32 or q.device != self.device or not q.is_contiguous():
            q = q.to(device=self.device, dtype=torch.float32).contiguous()
        if q.numel() != self.T * self.slab_stride:
            raise ValueError("q must be [T, stride]")
        face = None
        if face_variant is not None:
            P = FACE_PLANES[face_variant]
            face = face_out if face_out is not None else torch.empty((self.T, P, 15, 4), dtype=torch.float32, device=self.device)
            if face.dtype != torch.float32 or face.numel() != self.T * P * 60 or not face.is_contiguous():
                raise ValueError(f"face_out must be a contiguous float32 [T,{P},15,4] tensor")
        if traj is not None and (traj.dtype != torch.uint8 or traj.numel() != self.T * TRAJ_BYTES or not traj.is_contiguous()):
            raise ValueError("traj must be a contiguous uint8 [T,32] tensor")
        if choice_out is not None and (choice_out.dtype != torch.int32 or choice_out.numel() != self.T):
            raise ValueError("choice_out must be int32 [T]")
        pp = self._pp
        check(self.lib.ddz_policy_step_slab(self._h, _p(q), float(epsilon), pp["counts"], pp["rows"], pp["ids"],
                                            self.slab_stride, 1 if auto_reset else 0, pp["done"], pp["reward"],
                                            pp["illegal"], _p(traj), _p(choice_out),
                                            int(face_variant) if face_variant is not None else 0, _p(face),
                                            _stream(self.device)))
        self._legal_fresh, self._slab_fresh, self._csr_fresh = False, True, False
        return self.done, self.reward, self.illegal, face

    def slab_to_csr(self, rows_per_table=64):
        """The slab lists as CSR (offsets int32[T+1], rows int8[cap,16], ids int32[cap] | None) -- the layout legal()
        returns and a ragged NN forward consumes -- by two small launches; list indices are the same in both layouts, so
        select(q_csr) feeds step_slab(CHOICE).  cap = T * rows_per_table rows (mean list: 5.6 rows under a random
        policy; the first lead of a game: ~73): a fuller list raises status bit 1 and is truncated."""
        if not self._slab_fresh:
            self.legal_slab()
        cap = self.T * int(rows_per_table)
        if getattr(self, "_csr_cap", 0) != cap:
            self.csr_rows = torch.empty((cap, ROW), dtype=torch.int8, device=self.device)
            self.csr_ids = torch.empty(cap, dtype=torch.int32, device=self.device) if self.ids is not None else None
            self._csr_cap = cap
        check(self.lib.ddz_slab_to_csr(self._h, self._pp["counts"], self._pp["rows"], self._pp["ids"], self.slab_stride,
                                       self._pp["offsets"], _p(self.csr_rows), _p(self.csr_ids), cap, _stream(self.device)))
        self._csr_fresh = True    # select(q_csr) may use self.offsets as they are; the slab lists stay valid for step_slab
        return self.offsets, self.csr_rows, self.csr_ids

    def rollout_random(self, n_iters, traj=None):
        """n_iters lock-step iterations of {legal list, step_random(auto_reset)}
        (game.py:169-181 with envi.py:79-85), one kernel launch each.  The lists of the last
        iteration's *pre-step* states are left in the slab layout (counts, slab_rows())."""
        if self.slab_stride < MAX_LEGAL_PER_TABLE:
            raise ValueError(f"rollout needs row_capacity >= {MAX_LEGAL_PER_TABLE} * n_tables")
        if traj is not None and (traj.dtype != torch.uint8 or not traj.is_contiguous()
                                 or traj.numel() != n_iters * self.T * TRAJ_BYTES):
            raise ValueError("traj must be a contiguous uint8 [n_iters,T,32] tensor")
        check(self.lib.ddz_rollout_random(self._h, int(n_iters), _p(self.counts), _p(self.rows),
                                          _p(self.ids), self.slab_stride, _p(self._stats), _p(traj),
                                          _stream(self.device)))
        self._legal_fresh = self._slab_fresh = self._csr_fresh = False

    def rollout_random_csr(self, n_iters, traj=None, batch=None):
        """The same loop with packed CSR lists (offsets/rows/ids as legal() returns them: afterwards they hold the lists
        of the last iteration's pre-step states).  Same states and trajectories as rollout_random.
        batch (default: as many iterations as CSR_STAGING_BYTES = 16 GiB of staging hold -- 32 at 65,536 tables, 512 at 4096;
        the device has 288 GB -- at most 512): the lists of a batch of iterations are staged as slabs by ONE rollout launch and
        compacted to CSR by two more (ddz_rollout_random_csr_staged) -- no launch per iteration, and the longer the batch the
        less the launch's prologue weighs (tools/csr_batch_probe.py: 65,536 tables 1.95 G steps/s at 4 iterations per batch,
        2.81 G at 32); batch=0: the one-launch-per-iteration form (ddz_rollout_random_csr: CSR bases depend on every table,
        so each iteration waits for the scan of the one before)."""
        if traj is not None and (traj.dtype != torch.uint8 or not traj.is_contiguous()
                                 or traj.numel() != n_iters * self.T * TRAJ_BYTES):
            raise ValueError("traj must be a contiguous uint8 [n_iters,T,32] tensor")
        if batch is None:
            per = self.T * MAX_LEGAL_PER_TABLE * (20 if self.ids is not None else 16)
            batch = max(2, min(512, CSR_STAGING_BYTES // per))
        if batch:
            batch = int(min(batch, max(1, n_iters)))
            need = self.lib.ddz_rollout_csr_staging_bytes(self.T, batch, int(self.ids is not None))
            if getattr(self, "_staging", None) is None or self._staging.numel() < need:
                self._staging = torch.empty(need, dtype=torch.uint8, device=self.device)
            check(self.lib.ddz_rollout_random_csr_staged(self._h, int(n_iters), batch, _p(self._staging), self._staging.numel(),
                                                         _p(self.offsets), _p(self.rows), _p(self.ids), self.cap, _p(traj),
                                                         _stream(self.device)))
        else:
            check(self.lib.ddz_rollout_random_csr(self._h, int(n_iters), _p(self.offsets), _p(self.rows),
                                                  _p(self.ids), self.cap, _p(traj), _stream(self.device)))
        self._legal_fresh = self._slab_fresh = self._csr_fresh = False

    def rollout_random_timed(self, n_iters):
        """Same loop between two hipEvents; returns the elapsed ms of the n_iters launches.
        Synchronises; measurement aid for bench.py."""
        ms = (C.c_double * 2)()
        check(self.lib.ddz_rollout_random_timed(self._h, int(n_iters), _p(self.counts), _p(self.rows),
                                                _p(self.ids), self.slab_stride, ms,
                                                _stream(self.device)))
        self._legal_fresh = self._slab_fresh = self._csr_fresh = False
        return ms[0]

    def stats(self):
        """{plies, episodes, legal_rows, lord_wins, up_wins, down_wins} accumulated so far (the
        per-role win counts of Game.compete, game.py:258-290); host sync."""
        check(self.lib.ddz_read_stats(self._h, _p(self._stats), _stream(self.device)))
        p, e, l, w, u, dn = self._stats.tolist()[:6]
        return {"plies": p, "episodes": e, "legal_rows": l, "lord_wins": w, "up_wins": u, "down_wins": dn}

    def status(self):
        """device status word (0 = healthy); host sync."""
        out = C.c_int32(0)
        check(self.lib.ddz_status(self._h, C.byref(out), _stream(self.device)))
        return out.value

    # ---- checkpoint (the whole env is one byte tensor) ----
    def state_export(self):
        return self.state.clone()

    def state_import(self, state):
        if state.numel() != self.state.numel():
            raise ValueError("state size mismatch")
        self.state.copy_(state.to(self.device).view(-1))
        check(self.lib.ddz_invalidate(self._h))
        self._legal_fresh = self._slab_fresh = self._csr_fresh = False


def rows_to_onehot(rows):
    """batch_arr2onehot (envi.py:139-146) on device: int8 [n,16] -> f32 [n,15,4]."""
    L = _lib.lib()
    dev = _require_gpu(rows.device)
    rows = rows.to(torch.int8).contiguous()
    n = rows.shape[0]
    out = torch.empty((n, 15, 4), dtype=torch.float32, device=dev)
    if n:
        check(L.ddz_rows_to_onehot(dev.index, _p(rows), n, _p(out), _stream(dev)))
    return out


def state_prob(known60, size1, size2, device="cuda:0"):
    """get_state_prob_manual(known60, size1, size2) (server/core.py:26-33) for a batch: known60 [n,60] thermometers
    of own cards + cards played, size1 / size2 [n] = cards left of the next / next-but-one player.  Returns f32
    [n,2,15,4] on the device (prob planes spec v1: the last two planes of `face`)."""
    L = _lib.lib()
    dev = _require_gpu(device)
    k = torch.as_tensor(known60).reshape(-1, 60)
    k = (k != 0).to(device=dev, dtype=torch.uint8).contiguous()
    n = k.shape[0]
    sizes = torch.stack([torch.as_tensor(size1).reshape(n), torch.as_tensor(size2).reshape(n)], 1)
    sizes = sizes.to(device=dev, dtype=torch.int32).contiguous()
    out = torch.empty((n, 2, 15, 4), dtype=torch.float32, device=dev)
    if n:
        check(L.ddz_state_prob(dev.index, _p(k), _p(sizes), n, _p(out), _stream(dev)))
    return out


def q_features(face, wf, bias, acnt, y):
    """ddz_q_features: first layer of the ragged Q forward per (table, rank, count) from `face` f32 [T,P,15,4] into
    y f32 [15,5,T,K] (K >= 256; columns >= 256 and counts 2..4 of the joker ranks are left alone).
    dqn_glue.FactorisedQ.tables drives it."""
    L = _lib.lib()
    dev = _require_gpu(face.device)
    T, P = int(face.shape[0]), int(face.shape[1])
    if face.dtype != torch.float32 or tuple(face.shape[2:]) != (15, 4) or not face.is_contiguous():
        raise ValueError("face must be a contiguous float32 [T,P,15,4] tensor")
    if y.dtype != torch.float32 or tuple(y.shape[:3]) != (15, 5, T) or not y.is_contiguous() or y.device != dev:
        raise ValueError("y must be a contiguous float32 [15,5,T,K] tensor on the same device")
    for w, n in ((wf, P * 4 * 1024), (bias, 1024), (acnt, 5 * 4 * 256)):
        if w.dtype != torch.float32 or w.numel() != n or not w.is_contiguous() or w.device != dev:
            raise ValueError("weight tables must be contiguous float32 device tensors: wf [P*4,1024], bias [1024], acnt [5,4,256]")
    check(L.ddz_q_features(dev.index, _p(face), T, P, _p(wf), _p(bias), _p(acnt), _p(y), int(y.shape[3]), _stream(dev)))
    return y


def q_features_packed(face, wf, bias, acnt, row_index, rank_row0, y):
    """ddz_q_features_packed: the same first layer, written only for the (rank, count, table) rows a legal move can use:
    row_index int32 [T,64] + rank_row0 (16 host ints) as dqn_glue.FactorisedQ.pack builds them, y f32 [>= n_rows, K]."""
    L = _lib.lib()
    dev = _require_gpu(face.device)
    T, P = int(face.shape[0]), int(face.shape[1])
    n_rows = int(rank_row0[15])
    if face.dtype != torch.float32 or tuple(face.shape[2:]) != (15, 4) or not face.is_contiguous():
        raise ValueError("face must be a contiguous float32 [T,P,15,4] tensor")
    if y.dtype != torch.float32 or y.dim() != 2 or y.shape[0] < n_rows or not y.is_contiguous() or y.device != dev:
        raise ValueError("y must be a contiguous float32 [>= n_rows, K] tensor on the same device")
    if row_index.dtype != torch.int32 or tuple(row_index.shape) != (T, 64) or not row_index.is_contiguous() or row_index.device != dev:
        raise ValueError("row_index must be a contiguous int32 [T,64] tensor on the same device")
    for w, n in ((wf, P * 4 * 1024), (bias, 1024), (acnt, 5 * 4 * 256)):
        if w.dtype != torch.float32 or w.numel() != n or not w.is_contiguous() or w.device != dev:
            raise ValueError("weight tables must be contiguous float32 device tensors: wf [P*4,1024], bias [1024], acnt [5,4,256]")
    r0 = (C.c_int64 * 15)(*[int(x) for x in rank_row0[:15]])
    check(L.ddz_q_features_packed(dev.index, _p(face), T, P, _p(wf), _p(bias), _p(acnt), _p(row_index), r0, n_rows, _p(y),
                                  int(y.shape[1]), _stream(dev)))
    return y


def q_need_scratch_bytes(n_tables):
    return int(_lib.lib().ddz_q_need_scratch_bytes(int(n_tables)))


def q_features_needed(face, wf, bias, acnt, row_index, y0, dy):
    """ddz_q_features_needed: first layer into y0 f32 [T, 15 * 256] (count 0 of every rank) and dy f32 [rows, 256]
    (Y[count] - Y[0] at the row of every needed (table, rank, count): row_index from BatchedEnv.q_need)."""
    L = _lib.lib()
    dev = _require_gpu(face.device)
    T, P = int(face.shape[0]), int(face.shape[1])
    if face.dtype != torch.float32 or tuple(face.shape[2:]) != (15, 4) or not face.is_contiguous():
        raise ValueError("face must be a contiguous float32 [T,P,15,4] tensor")
    if y0 is not None and (y0.dtype != torch.float32 or tuple(y0.shape) != (T, 15 * 256) or not y0.is_contiguous() or y0.device != dev):
        raise ValueError("y0 must be a contiguous float32 [T, 3840] tensor on the same device (or None: dy alone)")
    if dy.dtype != torch.float32 or dy.dim() != 2 or dy.shape[1] != 256 or not dy.is_contiguous() or dy.device != dev:
        raise ValueError("dy must be a contiguous float32 [rows, 256] tensor on the same device")
    if row_index.dtype != torch.int32 or tuple(row_index.shape) != (T, 64) or not row_index.is_contiguous() or row_index.device != dev:
        raise ValueError("row_index must be a contiguous int32 [T,64] tensor on the same device")
    for w, n in ((wf, P * 4 * 1024), (bias, 1024), (acnt, 5 * 4 * 256)):
        if w.dtype != torch.float32 or w.numel() != n or not w.is_contiguous() or w.device != dev:
            raise ValueError("weight tables must be contiguous float32 device tensors: wf [P*4,1024], bias [1024], acnt [5,4,256]")
    check(L.ddz_q_features_needed(dev.index, _p(face), T, P, _p(wf), _p(bias), _p(acnt), _p(row_index), _p(y0), _p(dy),
                                  int(dy.shape[0]), _stream(dev)))


def q_fc1_dense(a, w, c):
    """ddz_q_fc1_dense: c f32 [n,256] += a f32 [n,k] @ w f32 [k,256] on the fp32 matrix cores (exact f32; k % 16 == 0)."""
    L = _lib.lib()
    dev = _require_gpu(a.device)
    n, k = int(a.shape[0]), int(a.shape[1])
    for x, shp in ((a, (n, k)), (w, (k, 256)), (c, (n, 256))):
        if x.dtype != torch.float32 or tuple(x.shape) != shp or not x.is_contiguous() or x.device != dev:
            raise ValueError("a [n,k], w [k,256], c [n,256]: contiguous float32 tensors on one device")
    check(L.ddz_q_fc1_dense(dev.index, _p(a), n, k, _p(w), _p(c), _stream(dev)))
    return c


def q_fc1_rows(dy, seg, row_cnt, w2, z, d):
    """ddz_q_fc1_rows: d[row] = dy[row] @ w2[rank of the row] + z[rank][row_cnt[row]] for the rows / rank segments of seg
    (device int32 [40]); z f32 [15,5,256]."""
    L = _lib.lib()
    dev = _require_gpu(dy.device)
    n = int(dy.shape[0])
    for x, shp in ((dy, (n, 256)), (d, (n, 256)), (w2, (15, 256, 256))):
        if x.dtype != torch.float32 or tuple(x.shape) != shp or not x.is_contiguous() or x.device != dev:
            raise ValueError("dy / d [rows,256], w2 [15,256,256]: contiguous float32 tensors on one device")
    if seg.dtype != torch.int32 or seg.numel() < 40 or seg.device != dev:
        raise ValueError("seg must be int32 [40] on the same device")
    if (row_cnt is None) != (z is None):
        raise ValueError("row_cnt and z: both or neither")
    if z is not None and (row_cnt.dtype != torch.uint8 or row_cnt.numel() < n or row_cnt.device != dev or z.dtype != torch.float32
                          or z.numel() != 75 * 256 or not z.is_contiguous() or z.device != dev):
        raise ValueError("row_cnt must be uint8 [rows], z float32 [15,5,256], on the same device")
    check(L.ddz_q_fc1_rows(dev.index, _p(dy), _p(seg), _p(row_cnt), _p(w2), _p(z), _p(d), n, _stream(dev)))
    return d


def q_shared_ws_bytes():
    return int(_lib.lib().ddz_q_shared_ws_bytes())


def q_features_rows(face, wf, bias, rep, seg, ys, mz=None, g=None):
    """ddz_q_features_rows: ys f32 [rows,256] = the first layer (count 0) of the face column of every shared row (rep: row ->
    instance 16 t + r, from BatchedEnv.q_shared_rows); face f32 [T,6,15,4]."""
    L = _lib.lib()
    dev = _require_gpu(face.device)
    T, P = int(face.shape[0]), int(face.shape[1])
    if face.dtype != torch.float32 or tuple(face.shape[1:]) != (6, 15, 4) or not face.is_contiguous():
        raise ValueError("face must be a contiguous float32 [T,6,15,4] tensor (EnvCooperationSimplify)")
    n = int(ys.shape[0])
    if ys.dtype != torch.float32 or ys.dim() != 2 or ys.shape[1] not in (256, 288) or not ys.is_contiguous() or ys.device != dev:
        raise ValueError("ys must be a contiguous float32 [rows,256] (or [rows,288]: + the column values) tensor on the same device")
    if rep.dtype != torch.int32 or rep.numel() < n or rep.device != dev or seg.dtype != torch.int32 or seg.numel() < 40 or seg.device != dev:
        raise ValueError("rep must be int32 [rows], seg int32 [40], on the same device")
    for w, k in ((wf, P * 4 * 1024), (bias, 1024)):
        if w.dtype != torch.float32 or w.numel() != k or not w.is_contiguous() or w.device != dev:
            raise ValueError("weight tables must be contiguous float32 device tensors: wf [P*4,1024], bias [1024]")
    if (mz is None) != (g is None):
        raise ValueError("mz and g: both or neither")
    if mz is not None and (mz.dtype != torch.float32 or tuple(mz.shape) != (P * 60, 256) or not mz.is_contiguous() or mz.device != dev
                           or g.dtype != torch.float32 or tuple(g.shape) != (n, 256) or not g.is_contiguous() or g.device != dev):
        raise ValueError("mz must be float32 [60 P, 256], g float32 [rows, 256], contiguous, on the same device")
    check(L.ddz_q_features_rows(dev.index, _p(face), T, P, _p(wf), _p(bias), _p(rep), _p(seg), _p(ys), int(ys.shape[1]), n, _p(mz), _p(g),
                                _stream(dev)))
    return ys


def q_fc1_rows_k(y, seg, w2k, g, accumulate=False):
    """ddz_q_fc1_rows_k: g[row] (+)= y[row] @ w2k[rank of the row] for the rows / rank segments of seg (device int32 [40]);
    y f32 [rows,k], w2k f32 [15,k,256], k a multiple of 16."""
    L = _lib.lib()
    dev = _require_gpu(y.device)
    n, k = int(y.shape[0]), int(y.shape[1])
    for x, shp in ((y, (n, k)), (g, (n, 256)), (w2k, (15, k, 256))):
        if x.dtype != torch.float32 or tuple(x.shape) != shp or not x.is_contiguous() or x.device != dev:
            raise ValueError("y [rows,k], g [rows,256], w2k [15,k,256]: contiguous float32 tensors on one device")
    if seg.dtype != torch.int32 or seg.numel() < 40 or seg.device != dev:
        raise ValueError("seg must be int32 [40] on the same device")
    check(L.ddz_q_fc1_rows_k(dev.index, _p(y), k, _p(seg), _p(w2k), _p(g), n, 1 if accumulate else 0, _stream(dev)))
    return g


def q_shared_need_ws_bytes(shared_row_capacity):
    n = int(_lib.lib().ddz_q_shared_need_ws_bytes(int(shared_row_capacity)))
    if n < 0:
        raise ValueError("the shared row capacity must be a positive multiple of the fc1 tile")
    return n


def q_features_drows(face, wf, bias, acnt, rep, drep, dseg, dy):
    """ddz_q_features_drows: dy f32 [rows,256] = Y[c] - Y[0] of the column of every shared D row (drep: D row -> 4 * shared row
    + c - 1, rep: shared row -> instance; from BatchedEnv.q_shared_need / q_shared_rows); face f32 [T,6,15,4]."""
    L = _lib.lib()
    dev = _require_gpu(face.device)
    T, P = int(face.shape[0]), int(face.shape[1])
    if face.dtype != torch.float32 or tuple(face.shape[1:]) != (6, 15, 4) or not face.is_contiguous():
        raise ValueError("face must be a contiguous float32 [T,6,15,4] tensor (EnvCooperationSimplify)")
    n = int(dy.shape[0])
    if dy.dtype != torch.float32 or dy.dim() != 2 or dy.shape[1] != 256 or not dy.is_contiguous() or dy.device != dev:
        raise ValueError("dy must be a contiguous float32 [rows,256] tensor on the same device")
    for x in (rep, drep, dseg):
        if x.dtype != torch.int32 or x.device != dev or not x.is_contiguous():
            raise ValueError("rep / drep / dseg must be contiguous int32 tensors on the same device")
    if drep.numel() < n or dseg.numel() < 40:
        raise ValueError("drep must hold a slot per row of dy, dseg 40 ints")
    for w, k in ((wf, P * 4 * 1024), (bias, 1024), (acnt, 5 * 4 * 256)):
        if w.dtype != torch.float32 or w.numel() != k or not w.is_contiguous() or w.device != dev:
            raise ValueError("weight tables must be contiguous float32 device tensors: wf [P*4,1024], bias [1024], acnt [5,4,256]")
    check(L.ddz_q_features_drows(dev.index, _p(face), T, P, _p(wf), _p(bias), _p(acnt), _p(rep), rep.numel(), _p(drep), _p(dseg),
                                 _p(dy), n, _stream(dev)))
    return dy


def q_gather_h0(g, rows, h0, base=None):
    """ddz_q_gather_h0: h0 f32 [T,256] (+)= sum_r g[rows[t, r]] (rank order); g f32 [g_rows,256], rows int32 [T,16]; with
    base f32 [256]: h0 = base + the sum."""
    L = _lib.lib()
    dev = _require_gpu(g.device)
    T = int(h0.shape[0])
    for x, shp in ((g, (int(g.shape[0]), 256)), (h0, (T, 256))):
        if x.dtype != torch.float32 or tuple(x.shape) != shp or not x.is_contiguous() or x.device != dev:
            raise ValueError("g [g_rows,256], h0 [T,256]: contiguous float32 tensors on one device")
    if rows.dtype != torch.int32 or tuple(rows.shape) != (T, 16) or not rows.is_contiguous() or rows.device != dev:
        raise ValueError("rows must be a contiguous int32 [T,16] tensor on the same device")
    if base is not None and (base.dtype != torch.float32 or base.numel() != 256 or not base.is_contiguous() or base.device != dev):
        raise ValueError("base must be a contiguous float32 [256] tensor on the same device")
    check(L.ddz_q_gather_h0(dev.index, _p(g), int(g.shape[0]), _p(rows), T, _p(base), _p(h0), _stream(dev)))
    return h0


def action_table(device="cuda:0", native_joker_kickers=False):
    """The canonical action table on the device: int8 [n_actions, 16] = counts[15] + category, action id =
    row index (the order of card.py:34-159 get_action_space(); + the 24 joker-kicker rows if asked for)."""
    L = _lib.lib(jk=native_joker_kickers)
    dev = _require_gpu(device)
    rows = torch.empty((L.ddz_num_actions(), ROW), dtype=torch.int8, device=dev)
    check(L.ddz_action_table(dev.index, _p(rows), _stream(dev)))
    return rows


TRAJ_PACKED_BYTES = 8


def pack_trajectory(traj, native_joker_kickers=False):
    """uint8 [..., 32] trajectory records -> uint8 [..., 8] (include/ddz_env.h ddz_pack_trajectory): the action
    as its canonical id.  What dist.gather_trajectories(compact=True) sends over xGMI."""
    L = _lib.lib(jk=native_joker_kickers)
    dev = _require_gpu(traj.device)
    if traj.dtype != torch.uint8 or traj.shape[-1] != TRAJ_BYTES:
        raise ValueError("traj must be uint8 [..., 32]")
    traj = traj.contiguous()
    n = traj.numel() // TRAJ_BYTES
    out = torch.empty(traj.shape[:-1] + (TRAJ_PACKED_BYTES,), dtype=torch.uint8, device=dev)
    check(L.ddz_pack_trajectory(dev.index, _p(traj), n, _p(out), _stream(dev)))
    return out


def get_moves(hands, lasts, want_ids=True, row_capacity=None, native_joker_kickers=False):
    """Batched r.get_moves(hand15, last15) (envi.py:111): hands/lasts int8 [n,15|16] on the
    GPU.  Returns (offsets[n+1] i32, rows[total,16] i8, ids[total] i32 | None); one host sync
    to trim the outputs.  native_joker_kickers: see BatchedEnv."""
    L = _lib.lib(jk=native_joker_kickers)
    NUM_ACTIONS = L.ddz_num_actions()
    dev = _require_gpu(hands.device)

    def pad(x):
        x = x.to(device=dev, dtype=torch.int8)
        if x.shape[1] == 15:
            x = torch.nn.functional.pad(x, (0, 1))
        return x.contiguous()

    hands, lasts = pad(hands), pad(lasts)
    n = hands.shape[0]
    if lasts.shape[0] != n:
        raise ValueError("hands and lasts must have the same length")
    cap = int(row_capacity) if row_capacity is not None else min(n * NUM_ACTIONS, 0x7FFFFFFF)
    offsets = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    scratch = torch.zeros(L.ddz_scratch_bytes(n), dtype=torch.uint8, device=dev)

    def run(capacity):
        rows = torch.empty((max(capacity, 1), ROW), dtype=torch.int8, device=dev)
        ids = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev) if want_ids else None
        check(L.ddz_get_moves(dev.index, _p(hands), _p(lasts), n, _p(offsets), _p(rows), _p(ids),
                              capacity, _p(scratch), scratch.numel(), _stream(dev)))
        return rows, ids

    if row_capacity is None and n * NUM_ACTIONS > (1 << 22):
        run(0)  # sizes only: offsets are exact whatever the capacity
        cap = int(offsets[-1].item())
        scratch.zero_()
    rows, ids = run(cap)
    total = int(offsets[-1].item())
    if total > cap:
        raise DdzError(f"row capacity {cap} too small for {total} rows")
    bad = int(scratch.view(torch.int32)[(scratch.numel() - 256) // 4].item()) & 4
    if bad:
        raise ValueError("a `last` vector is not a combo of the action space")
    return offsets, rows[:total], (ids[:total] if want_ids else None)


def auto_choose(hands, lasts, left, role, want_stats=False):
    """RuleBasedModel.choose for n independent queries (what server/core.py:80-87 calls on a payload): hands / lasts
    int8 [n,15|16] (last all-zero = lead), left int [n,3] = cards left of role 0 up / 1 lord / 2 down, role int [n].
    Returns int32[n] canonical action ids (0 = pass, -2 = invalid query) (and int64 [n,2] {combinations, nodes})."""
    L = _lib.lib()
    dev = _require_gpu(hands.device)

    def pad(x):
        x = x.to(device=dev, dtype=torch.int8)
        if x.shape[1] == 15:
            x = torch.nn.functional.pad(x, (0, 1))
        return x.contiguous()

    hands, lasts = pad(hands), pad(lasts)
    n = hands.shape[0]
    info = torch.zeros((n, 4), dtype=torch.uint8, device=dev)
    info[:, :3] = torch.as_tensor(left, device=dev).reshape(n, 3).to(torch.uint8)
    info[:, 3] = torch.as_tensor(role, device=dev).reshape(n).to(torch.uint8)
    ids = torch.empty(n, dtype=torch.int32, device=dev)
    stats = torch.zeros((n, 2), dtype=torch.int64, device=dev) if want_stats else None
    check(L.ddz_auto_choose(dev.index, _p(hands), _p(lasts), _p(info), n, _p(ids), _p(stats), _stream(dev)))
    return (ids, stats) if want_stats else ids


def device_status(device="cuda:0"):
    """Status bits of the stateless rule-agent launches on this device since the last call (ddz_device_status; bit 3 = a
    cooperating wait hit its hang guard: do not trust those ids); host sync, clears the word."""
    L = _lib.lib()
    dev = _require_gpu(device)
    out = C.c_int32(0)
    check(L.ddz_device_status(dev.index, C.byref(out), _stream(dev)))
    return out.value


def cards_value(device="cuda:0"):
    """cards_value of rule_based/utils/evaluator.py:10-47 for every action id, float64 [13527] (device test hook)."""
    L = _lib.lib()
    dev = _require_gpu(device)
    out = torch.empty(NUM_ACTIONS, dtype=torch.int8, device=dev)
    check(L.ddz_debug_cards_value(dev.index, _p(out), _stream(dev)))
    return out.to(torch.float64) / 2.0


def get_moves_slab(hands, lasts, want_ids=True, native_joker_kickers=False, out=None):
    """Batched r.get_moves(hand15, last15) in the slab layout, ONE launch and no host sync: returns (counts int32 [n],
    rows int8 [n, 512, 16], ids int32 [n, 512] | None, status int32 [1]); query i's moves are rows[i, :counts[i]] in
    ascending canonical id (pass first when following); status bit 2 = some `last` was no combo (its list is empty).
    `out` = a previous result to write into (no allocation)."""
    L = _lib.lib(jk=native_joker_kickers)
    dev = _require_gpu(hands.device)

    def pad(x):
        x = x.to(device=dev, dtype=torch.int8)
        if x.shape[1] == 15:
            x = torch.nn.functional.pad(x, (0, 1))
        return x.contiguous()

    hands, lasts = pad(hands), pad(lasts)
    n = hands.shape[0]
    if lasts.shape[0] != n:
        raise ValueError("hands and lasts must have the same length")
    if out is None:
        counts = torch.empty(n, dtype=torch.int32, device=dev)
        rows = torch.empty((n, MAX_LEGAL_PER_TABLE, ROW), dtype=torch.int8, device=dev)
        ids = torch.empty((n, MAX_LEGAL_PER_TABLE), dtype=torch.int32, device=dev) if want_ids else None
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    else:
        counts, rows, ids, status = out
        status.zero_()
    check(L.ddz_get_moves_slab(dev.index, _p(hands), _p(lasts), n, _p(counts), _p(rows), _p(ids), MAX_LEGAL_PER_TABLE,
                               _p(status), _stream(dev)))
    return counts, rows, ids, status
