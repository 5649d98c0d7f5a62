"""Stateless observation / legal moves from the reference's serving payload, and the payloads of live tables
(state_to_payloads <-> payloads_to_state).

The reference's HTTP predictor receives, per request, a dict
    {role_id, cur_cards, history{0,1,2}, left{0,1,2}, last_taken{0,1,2}}      (server/client.py:6-25)
with card lists as rank values 3..17, and derives from it the EnvCooperationSimplify `face`
(server/core.py:44-54, prob planes via get_state_prob_manual :26-33) and the legal moves
against last_taken[(role-1)%3] or, if empty, last_taken[(role-2)%3] (server/core.py:56-67).
Here a batch of payloads is packed into the engine's state rows on the host (a few hundred
bytes per request) and handed to the same kernels the batched env uses (ddz_observe,
ddz_get_moves), so a served observation is bit-identical to the one of a live table.
"""
import numpy as np
import torch

from .engine import (BatchedEnv, F_HAND0, F_HIST0, F_META, F_RECENT0, F_TAKEN, NFIELDS, ROW, get_moves)


def _counts(cards):
    out = np.zeros(15, np.uint8)
    for c in cards:
        out[int(c) - 3] += 1                      # envi.py:132-137 cards2arr
    return out


def _get(d, k):
    return d[k] if k in d else d[str(k)]          # JSON turns the integer role keys into strings


def payloads_to_state(payloads):
    """uint8 [n, 11, 16] state rows of n payloads (other players' hands are unknown: only
    their sizes are set, which is all `face` reads of them)."""
    st = np.zeros((len(payloads), NFIELDS, ROW), np.uint8)
    for i, p in enumerate(payloads):
        role = int(p["role_id"])
        st[i, F_HAND0 + role, :15] = _counts(p["cur_cards"])
        for r in range(3):
            st[i, F_HAND0 + r, 15] = int(_get(p["left"], r))
            st[i, F_HIST0 + r, :15] = _counts(_get(p["history"], r))
            st[i, F_RECENT0 + r, :15] = _counts(_get(p["last_taken"], r))
        st[i, F_TAKEN, :15] = st[i, F_HIST0:F_HIST0 + 3, :15].sum(0)   # server/core.py:41 taken = h0 + h1 + h2
        st[i, F_META, 0] = role
        st[i, F_META, 2] = 0xFF
        st[i, F_META, 6] = 1
    return st


def state_to_payloads(state):
    """The inverse of payloads_to_state: state rows (uint8 [n,11,16] -- a numpy array, a tensor, or a BatchedEnv, whose
    state is copied to the host) -> the list of n serving payloads {role_id, cur_cards, history, left, last_taken} the
    reference's HTTP predictor takes (server/client.py:6-25: card lists as rank values 3..17, dicts keyed by role 0 up /
    1 lord / 2 down), one per table, as seen by the table's ACTOR (only its own hand is in a payload).  What a caller
    that plays on the batched engine POSTs to a reference-style server."""
    if isinstance(state, BatchedEnv):
        state = state.state
    if torch.is_tensor(state):
        state = state.detach().cpu().numpy()
    st = np.asarray(state, np.uint8).reshape(-1, NFIELDS, ROW)
    ranks = np.arange(3, 18)

    def cards(row):
        return [int(x) for x in np.repeat(ranks, row[:15].astype(int))]   # envi.py:118-130 arr2cards

    out = []
    for s in st:
        role = int(s[F_META, 0])
        out.append({"role_id": role, "cur_cards": cards(s[F_HAND0 + role]),
                    "history": {r: cards(s[F_HIST0 + r]) for r in range(3)},
                    "left": {r: int(s[F_HAND0 + r, 15]) for r in range(3)},
                    "last_taken": {r: cards(s[F_RECENT0 + r]) for r in range(3)}})
    return out


class BatchedPredictorInputs:
    """face / valid_actions of server/core.py's Predictor for a batch of payloads."""

    def __init__(self, device="cuda:0", variant=3):
        self.device = torch.device(device)
        self.variant = variant
        self._env = None

    def face(self, payloads):
        """f32 [n, P, 15, 4] on the device (server/core.py:44-54 for variant 3)."""
        n = len(payloads)
        if self._env is None or self._env.T != n:
            self._env = BatchedEnv(n, seed=0, device=self.device, row_capacity=max(512 * n, 512), want_ids=False)
        self._env.state_import(torch.from_numpy(payloads_to_state(payloads)).view(-1))
        return self._env.observe(self.variant)

    def valid_actions(self, payloads):
        """(last, offsets, rows): the combo each request has to beat as a rank list
        (server/core.py:57-60) and the CSR list of its legal moves (int8 rows, counts + category)."""
        hands = np.zeros((len(payloads), ROW), np.int8)
        lasts = np.zeros((len(payloads), ROW), np.int8)
        back = []
        for i, p in enumerate(payloads):
            role = int(p["role_id"])
            last = _get(p["last_taken"], (role + 2) % 3) or _get(p["last_taken"], (role + 1) % 3)
            back.append(list(last))
            hands[i, :15] = _counts(p["cur_cards"])
            lasts[i, :15] = _counts(last)
        offsets, rows, _ = get_moves(torch.from_numpy(hands).to(self.device), torch.from_numpy(lasts).to(self.device),
                                     want_ids=False)
        return back, offsets, rows
