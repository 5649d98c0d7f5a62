"""Drop-in for the reference's envi.py: the same class names, methods, shapes and dtypes
(`Env`, `EnvComplicated`, `EnvCooperation`, `EnvCooperationSimplify`; envi.py:16-217), backed
by a one-table BatchedEnv on the MI355X instead of the absent pybind modules `env` / `r`
(envi.py:10-13).  This is the N = 1 compatibility view for the existing game.py / dqn.py
loop; throughput comes from BatchedEnv, which these classes wrap.

Differences that cannot be avoided (documented in DESIGN.md 4): the deal / RNG (spec v2) and the two
probability planes of `face` (spec v1) are this repo's definitions because the native code that defined
them is not in the reference; step_auto plays the reference's Python rule agent (rule_based_model.py) over
"decomposer spec v1" in place of the absent native decomposition functions.
"""
import collections
import random

import numpy as np
import torch

from . import config as conf
from .engine import (BatchedEnv, F_HAND0, F_HIST0, F_META, F_RECENT0, F_TAKEN, STEP_CHOICE, STEP_IDS, STEP_ROWS,
                     rows_to_onehot, state_prob)


class _OldCards(dict):
    """Env.old_cards (envi.py:27,65: the actor's hand before its move, as cards 3..17 -- read by the debug print only,
    envi.py:46): kept as the 15 rank counts of the state row, turned into the reference's card array when read."""

    def __getitem__(self, role):
        return Env.arr2cards(dict.__getitem__(self, role))

    def get(self, role, default=None):
        return self[role] if role in self else default

    def values(self):
        return [self[k] for k in self]

    def items(self):
        return [(k, self[k]) for k in self]


class Env:
    FACE_VARIANT = 0  # envi.py:87-96: [hand, taken, prob1, prob2]

    def __init__(self, debug=False, seed=None, device=None):
        self.device = torch.device(device) if device is not None else conf.DEVICE
        # `if seed:` in the reference (envi.py:18-21): seed 0 / None -> unseeded
        self._seed = int(seed) if seed else random.getrandbits(63)
        # host_mirror: the table's 176 state bytes and its list size live in pinned host memory that the kernels write
        # directly; one ply = one launch (ddz_step_slab applies the move AND writes the next state's list) + ONE stream
        # synchronisation, no device -> host copy.  The selection travels the other way through pinned memory too.
        self._b = BatchedEnv(1, seed=self._seed, device=self.device, host_mirror=True)
        self._sel = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._row = torch.zeros((1, 16), dtype=torch.int8).pin_memory()
        self._ids = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._sel_np, self._row_np = self._sel.numpy(), self._row.numpy()   # (a numpy store costs 0.1 us, a tensor store 1.5)
        self._s = self._b.state.numpy().reshape(11, 16)          # live view of the state rows
        self._n = self._b.counts.numpy()                          # ... and of the size of the legal list
        self._m = self._s[F_META]                                 # the meta row (a view: always current)
        self._ply16 = self._m[4:6].view(np.uint16)                # its ply counter
        self._nlegal = 0
        self._views = None
        self.debug = debug
        self.old_cards = _OldCards()
        # lean call paths of the three per-ply calls (face / valid_actions / step): the library functions bound once, the
        # persistent buffers' pointers converted once (the generic BatchedEnv wrappers spend ~10 us per call on checks)
        import ctypes as C
        from .engine import FACE_PLANES, _raw_stream
        b = self._b
        self._C, self._L, self._h, self._di = C, b.lib, b._h, b.device.index
        self._P = FACE_PLANES[self.FACE_VARIANT]
        self._rows_ptr = b.rows.data_ptr()
        self._pp = b._pp
        self._sel_p, self._row_p, self._ids_p = (C.c_void_p(x.data_ptr()) for x in (self._sel, self._row, self._ids))
        self._raw = _raw_stream if _raw_stream is not None else (lambda i: torch.cuda.current_stream(self.device).cuda_stream)

    # ---- bookkeeping mirrors (envi.py:22-27): read from the state rows when asked for (fresh copies) ----
    @property
    def taken(self):
        return self._s[F_TAKEN, :15].astype(float)

    @property
    def left(self):
        return self._s[F_HAND0:F_HAND0 + 3, 15].astype(int)

    @property
    def history(self):
        return collections.defaultdict(lambda: np.zeros((15,)), {r: self._s[F_HIST0 + r, :15].astype(float) for r in range(3)})

    @property
    def recent_handout(self):
        return collections.defaultdict(lambda: np.zeros((15,)), {r: self._s[F_RECENT0 + r, :15].astype(int) for r in range(3)})

    @property
    def _meta(self):
        return self._s[F_META]

    @property
    def _hands(self):
        return self._s[F_HAND0:F_HAND0 + 3, :15]

    def _clear(self):
        self.old_cards = _OldCards()

    def _sync(self):
        rc = self._L.ddz_sync(self._di, self._C.c_void_p(self._raw(self._di)))   # the one wait of a ply
        if rc:
            from ._lib import check
            check(rc)
        self._nlegal = int(self._n[0])
        self._views = None

    def _observe_actions(self):
        """`face` and valid_actions() of the current state from ONE allocation and ONE library call (ddz_observe_actions:
        the launches of ddz_observe and ddz_rows_to_onehot): what an agent reads per ply (dqn.py:50-71).  Each of the two
        tensors is handed out once; a second read of the same property in the same ply computes a fresh one."""
        n, P = self._nlegal, self._P
        buf = torch.empty(((P + n) * 60,), dtype=torch.float32, device=self.device)
        p0 = buf.data_ptr()
        rc = self._L.ddz_observe_actions(self._h, self.FACE_VARIANT, self._C.c_void_p(p0), self._C.c_void_p(self._rows_ptr), n,
                                         self._C.c_void_p(p0 + P * 240), self._C.c_void_p(self._raw(self._di)))
        if rc:
            from ._lib import check
            check(rc)
        v = self._views = [buf[:P * 60].view(P, 15, 4), buf[P * 60:].view(n, 15, 4)]
        return v

    def reset(self):
        """envi.py:30-36: clears the adapter state; cards are dealt by prepare()."""
        self._clear()

    def prepare(self):
        """native shuffle + deal + lord selection (game.py:171); deal = spec v2."""
        self._b.reset()
        self._b.legal_slab()
        self._sync()

    # ---- native getters used by the callers ----
    def get_role_ID(self):
        return int(self._meta[0]) + 1  # 1-based (envi.py:64)

    def get_curr_handcards(self):
        return self.arr2cards(self._hands[int(self._meta[0])])

    def get_last_two_cards(self):
        role = int(self._meta[0])
        prev = self.arr2cards(self._s[F_RECENT0 + (role + 2) % 3, :15].astype(int))
        prevprev = self.arr2cards(self._s[F_RECENT0 + (role + 1) % 3, :15].astype(int))
        return [list(prev), list(prevprev)]

    def get_last_outcards(self):
        """native get_last_outcards(): the cards the actor has to beat, as ranks 3..17 -- empty when it leads.  Its one
        caller in the reference feeds it to get_mask as `last_cards` (rule_based/utils/utils.py:195-197), i.e. it is the
        `last` of valid_actions (envi.py:103-109): the previous player's handout, else the one before, else nothing."""
        prev, prevprev = self.get_last_two_cards()
        return np.asarray(prev if len(prev) else prevprev, dtype=int)

    def get_state_prob(self):
        """native get_state_prob() (envi.py:94): 120 floats, reshaped (2,15,4) by `face` -- prob planes spec v1."""
        return self._b.observe(0)[0, 2:4].reshape(120).cpu().numpy()

    def get_state_prob_manual(self, known60, size1, size2):
        """native get_state_prob_manual (server/core.py:26-33): the same planes from an explicit view of the game:
        known60 = flattened thermometer of own cards + cards played, size1 / size2 = cards left of the next two players."""
        return state_prob(np.asarray(known60).reshape(1, 60), [int(size1)], [int(size2)],
                          device=self.device)[0].reshape(120).cpu().numpy()

    # ---- stepping ----
    def _ply(self):
        return int(self._ply16[0])

    def _step(self, sel, mode):
        """one launch (apply + the next state's legal list) + ONE device -> host round trip (_sync): done / r are the
        meta row's, an action that was not in the legal list leaves the table untouched (same ply counter)"""
        m = self._m
        if m[1]:  # a finished table stays as it is (no auto-reset in this view)
            return 0, True
        before = int(self._ply16[0])
        b, pp = self._b, self._pp
        if not b._slab_fresh:
            b.legal_slab()
        sel_p = self._sel_p if sel is self._sel else self._row_p if sel is self._row else self._ids_p
        rc = self._L.ddz_step_slab(self._h, mode, sel_p, pp["counts"], pp["rows"], pp["ids"], b.slab_stride, 0, pp["done"],
                                   pp["reward"], pp["illegal"], None, self._C.c_void_p(self._raw(self._di)))
        if rc:
            from ._lib import check
            check(rc)
        b._legal_fresh, b._slab_fresh, b._csr_fresh = False, True, False    # (as BatchedEnv.step_slab)
        self._sync()
        if int(self._ply16[0]) == before:
            raise ValueError("illegal action for the current state")
        r = int(m[3])
        return (r - 256 if r > 127 else r), bool(m[1])

    def _apply(self, idx):
        role = int(self._m[0])
        self.old_cards[role] = self._s[F_HAND0 + role, :15].copy()
        self._sel_np[0] = idx
        r, done = self._step(self._sel, STEP_CHOICE)
        res = (r, done, None)
        if self.debug:
            print('role {} plays {}, left {}'.format(
                role, self.cards2str(self.arr2cards(self.recent_handout[role].astype(int))), self.left))
        return res

    def step_manual(self, onehot_cards):
        """envi.py:63-70: 15x4 thermometer -> (r, done, _); r -1 lord wins / +1 farmers.  The engine checks the cards
        against the legal list itself (DDZ_STEP_ROWS)."""
        role = int(self._m[0])
        self.old_cards[role] = self._s[F_HAND0 + role, :15].copy()
        self._row_np[0, :15] = self.onehot2arr(onehot_cards)
        r, done = self._step(self._row, STEP_ROWS)
        if self.debug:
            print('role {} plays {}, left {}'.format(
                role, self.cards2str(self.arr2cards(self.recent_handout[role].astype(int))), self.left))
        return r, done, None

    def step_auto(self):
        """envi.py:72-77: the rule-based opponent moves -> (cards, r, _).  The native step_auto is absent from the
        reference; this plays RuleBasedModel.choose (rule_based/utils/rule_based_model.py:43-101) on the device
        (decomposer spec v1)."""
        role = int(self._m[0])
        self.old_cards[role] = self._s[F_HAND0 + role, :15].copy()
        ids = self._b.auto_choose(0b111, out=self._ids)   # (the id lands in pinned memory; the step reads it from there)
        r, _ = self._step(ids, STEP_IDS)  # (an id the rule agent could not produce -- DDZ_AUTO_INVALID -- raises here)
        cards = self.arr2cards(self.recent_handout[role].astype(int))
        return cards, r, None

    def _legal(self):
        """(rows, n) of the current state: the list the last launch left in the table's slab, its size from the host
        mirror (no device round trip)"""
        return self._b.rows, self._nlegal

    def step_random(self):
        """envi.py:79-85 with Python's global `random` as in the reference."""
        return self._apply(random.randrange(self._legal()[1]))

    # ---- observations ----
    @property
    def face(self):
        v = self._views
        if v is None:
            v = self._observe_actions()
        if v[0] is not None:
            out, v[0] = v[0], None
            return out
        out = torch.empty((self._P, 15, 4), dtype=torch.float32, device=self.device)
        rc = self._L.ddz_observe(self._h, self.FACE_VARIANT, self._C.c_void_p(out.data_ptr()), self._C.c_void_p(self._raw(self._di)))
        if rc:
            from ._lib import check
            check(rc)
        return out

    def valid_actions(self, tensor=True):
        """envi.py:98-116: f32 [A,15,4] on the device, or a list of A int[15] arrays."""
        rows, n = self._legal()
        if tensor:
            v = self._views
            if v is None:
                v = self._observe_actions()
            if v[1] is not None:
                out, v[1] = v[1], None
                return out
            out = torch.empty((n, 15, 4), dtype=torch.float32, device=self.device)
            if n:
                rc = self._L.ddz_rows_to_onehot(self._di, self._C.c_void_p(self._rows_ptr), n, self._C.c_void_p(out.data_ptr()),
                                                self._C.c_void_p(self._raw(self._di)))
                if rc:
                    from ._lib import check
                    check(rc)
            return out
        return [a for a in rows[:n, :15].cpu().numpy().astype(int)]

    # ---- codecs (envi.py:118-161), host-side like the reference's ----
    @classmethod
    def arr2cards(cls, arr):
        arr = np.asarray(arr, dtype=int)
        return np.repeat(np.arange(3, 18), arr[:15])

    @classmethod
    def cards2arr(cls, cards):
        arr = np.zeros((15,), dtype=int)
        for card in cards:
            arr[int(card) - 3] += 1
        return arr

    @classmethod
    def batch_arr2onehot(cls, batch_arr):
        arr = np.asarray(batch_arr).reshape(len(batch_arr), 15)
        return (arr[:, :, None] > np.arange(4)[None, None, :]).astype(int)

    @classmethod
    def onehot2arr(cls, onehot_cards):
        if torch.is_tensor(onehot_cards):
            onehot_cards = onehot_cards.detach().cpu().numpy()
        return np.asarray(onehot_cards).reshape(15, 4).sum(axis=1).astype(int)

    def cards2str(self, cards):
        return [conf.DICT[int(i)] for i in cards]


class EnvComplicated(Env):
    FACE_VARIANT = 1  # envi.py:165-178: + history of (role-1, role, role+1)


class EnvCooperation(Env):
    FACE_VARIANT = 2  # envi.py:182-198: + history + recent handouts of (role-1, role-2)


class EnvCooperationSimplify(Env):
    FACE_VARIANT = 3  # envi.py:202-217: hand, taken, recent handouts, probs
