"""ctypes binding of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  See oracle/ddz_oracle.h for what is pinned against the reference
and what is "parity unpinned".
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NUM_ACTIONS = 13527
ROW = 16
NFIELDS = 11
TR_BYTES = 32
STEP_RANDOM, STEP_CHOICE, STEP_ROWS, STEP_IDS = 0, 1, 2, 3
PLANES = (4, 7, 9, 6)

_libs = {}
_jk = False  # which rule set lib() serves: False = card.py (13,527 rows), True = + the 24 joker-kicker rows


def build(force=False, jk=False):
    name = "libddz_oracle_jk.so" if jk else "libddz_oracle.so"
    so = os.path.join(HERE, name)
    deps = [os.path.join(HERE, f) for f in ("ddz_oracle.c", "ddz_auto_oracle.c", "ddz_oracle.h", "Makefile")]
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(p) > os.path.getmtime(so) for p in deps)
    if force or stale:
        subprocess.check_call(["make", "-C", HERE, name],
                              stdout=subprocess.DEVNULL)
    return so


class variant:
    """`with oracle.variant(jk=True):` -- every call inside uses the rule set with the 24 extra
    joker-kicker rows (oracle/ddz_oracle.c, DDZO_NATIVE_JOKER_KICKERS)."""

    def __init__(self, jk):
        self.jk = bool(jk)

    def __enter__(self):
        global _jk
        self.prev, _jk = _jk, self.jk
        return self

    def __exit__(self, *exc):
        global _jk
        _jk = self.prev


def num_actions():
    return int(lib().ddzo_num_actions())


def lib():
    if _jk not in _libs:
        L = C.CDLL(build(jk=_jk))
        L.ddzo_num_actions.restype = C.c_int
        p = C.c_void_p
        L.ddzo_init.restype = None
        L.ddzo_action_table.argtypes = [p, p]
        L.ddzo_beats.argtypes = [C.c_int, C.c_int]
        L.ddzo_beats.restype = C.c_int
        L.ddzo_lookup.argtypes = [p]
        L.ddzo_lookup.restype = C.c_int
        L.ddzo_legal.argtypes = [p, p, p, C.c_int]
        L.ddzo_legal.restype = C.c_int
        L.ddzo_philox4x32_10.argtypes = [p, p, p]
        L.ddzo_env_reset.argtypes = [p, C.c_int64, C.c_uint64, C.c_uint64, p]
        L.ddzo_env_legal.argtypes = [p, C.c_int64, p, p, p, C.c_int64]
        L.ddzo_env_legal.restype = C.c_int64
        L.ddzo_env_step.argtypes = [p, C.c_int64, C.c_uint64, C.c_uint64, C.c_int, p, p, p,
                                    C.c_int, p, p, p, p]
        L.ddzo_planes.argtypes = [C.c_int]
        L.ddzo_planes.restype = C.c_int
        L.ddzo_env_observe.argtypes = [p, C.c_int64, C.c_int, p]
        L.ddzo_rows_to_onehot.argtypes = [p, C.c_int64, p]
        L.ddzo_state_prob.argtypes = [p, C.c_int, C.c_int, p]
        L.ddzo_select.argtypes = [p, C.c_int64, C.c_uint64, C.c_uint64, p, p, C.c_double, p]
        L.ddzo_rollout_random.argtypes = [p, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, p, p]
        L.ddzo_rollout_random.restype = C.c_int64
        L.ddzo_cards_value_x2.argtypes = [p]
        for f in (L.ddzo_combinations_recursive, L.ddzo_combinations_nosplit):
            f.argtypes = [p, C.c_int, p, p, C.c_int64, p]
            f.restype = C.c_int64
        L.ddzo_auto_combinations.argtypes = [p, C.c_int, p, C.c_int64, p]
        L.ddzo_auto_combinations.restype = C.c_int64
        L.ddzo_auto_choose.argtypes = [p, p, p, C.c_int, p]
        L.ddzo_auto_choose.restype = C.c_int
        L.ddzo_env_auto_choose.argtypes = [p, C.c_int64, C.c_int, p, p]
        L.ddzo_init()
        L.ddzo_cards_value_x2(_ptr(np.zeros(NUM_ACTIONS, np.int16)))  # builds the rule agent's tables now, on this thread
        _libs[_jk] = L
    return _libs[_jk]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def action_table():
    rows = np.zeros((num_actions(), ROW), np.int8)
    info = np.zeros((num_actions(), 4), np.uint8)
    lib().ddzo_action_table(_ptr(rows), _ptr(info))
    return rows, info


def beats(a, b):
    return bool(lib().ddzo_beats(int(a), int(b)))


def lookup(counts15):
    c = np.ascontiguousarray(counts15, np.int8)
    return lib().ddzo_lookup(_ptr(c))


def legal(hand15, last15=None):
    """sorted canonical ids of get_mask(hand, action_space, last)."""
    h = np.ascontiguousarray(hand15, np.int8)
    l = None if last15 is None else np.ascontiguousarray(last15, np.int8)
    ids = np.zeros(num_actions(), np.int32)
    n = lib().ddzo_legal(_ptr(h), _ptr(l), _ptr(ids), len(ids))
    if n < 0:
        raise ValueError("last is not a combo of the action space (or bad hand)")
    return ids[:n].copy()


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().ddzo_philox4x32_10(_ptr(c), _ptr(k), _ptr(o))
    return o


class OracleEnv:
    """Batched CPU environment with the same state layout as the device engine."""

    def __init__(self, n_tables, seed=0, gid_base=0):
        self.T = int(n_tables)
        self.seed = int(seed)
        self.gid_base = int(gid_base)
        self.state = np.zeros(self.T * NFIELDS * ROW, np.uint8)
        self.offsets = np.zeros(self.T + 1, np.int32)
        self.cap = self.T * 512
        self.rows = np.zeros((self.cap, ROW), np.int8)
        self.ids = np.zeros(self.cap, np.int32)
        self.total = 0

    def field(self, f):
        return self.state.reshape(self.T, NFIELDS, ROW)[:, f]

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().ddzo_env_reset(_ptr(self.state), self.T, self.seed, self.gid_base, _ptr(m))

    def legal(self):
        self.total = lib().ddzo_env_legal(_ptr(self.state), self.T, _ptr(self.offsets),
                                          _ptr(self.rows), _ptr(self.ids), self.cap)
        assert self.total <= self.cap
        return self.offsets, self.rows[:self.total], self.ids[:self.total]

    def step(self, mode=STEP_RANDOM, sel=None, auto_reset=True, want_traj=False):
        done = np.zeros(self.T, np.uint8)
        reward = np.zeros(self.T, np.int8)
        illegal = np.zeros(self.T, np.uint8)
        traj = np.zeros((self.T, TR_BYTES), np.uint8) if want_traj else None
        if mode in (STEP_CHOICE, STEP_IDS):
            sel = np.ascontiguousarray(sel, np.int32)
        elif mode == STEP_ROWS:
            sel = np.ascontiguousarray(sel, np.int8)
        lib().ddzo_env_step(_ptr(self.state), self.T, self.seed, self.gid_base, mode, _ptr(sel),
                            _ptr(self.offsets), _ptr(self.rows), int(auto_reset), _ptr(done),
                            _ptr(reward), _ptr(illegal), _ptr(traj))
        return done, reward, illegal, traj

    def observe(self, variant):
        out = np.zeros((self.T, PLANES[variant], 15, 4), np.float32)
        lib().ddzo_env_observe(_ptr(self.state), self.T, variant, _ptr(out))
        return out

    def select(self, q, epsilon=0.0):
        q = np.ascontiguousarray(q, np.float32)
        out = np.zeros(self.T, np.int32)
        lib().ddzo_select(_ptr(self.state), self.T, self.seed, self.gid_base, _ptr(q), _ptr(self.offsets),
                          float(epsilon), _ptr(out))
        return out

    def auto_choose(self, auto_roles=0b101, want_stats=False):
        """Env.step_auto's choice (RuleBasedModel.choose) per table: action id, -1 where the actor is not a rule agent."""
        ids = np.zeros(self.T, np.int32)
        st = np.zeros(2, np.int64)
        lib().ddzo_env_auto_choose(_ptr(self.state), self.T, int(auto_roles), _ptr(ids), _ptr(st))
        return (ids, st) if want_stats else ids

    def rollout_random(self, n_iters):
        sl = C.c_int64(0)
        ep = C.c_int64(0)
        plies = lib().ddzo_rollout_random(_ptr(self.state), self.T, self.seed, self.gid_base,
                                          int(n_iters), C.byref(sl), C.byref(ep))
        return plies, sl.value, ep.value


def rollout_random_mt(env, n_iters, threads):
    """The same rollout with the tables split over `threads` host threads (tables are independent,
    game.py:28; RNG is keyed by the global table id, so the result equals the 1-thread run).
    ctypes releases the GIL around the C call."""
    from concurrent.futures import ThreadPoolExecutor
    lib().ddzo_init()
    T = env.T
    threads = max(1, min(int(threads), T))
    cuts = [T * i // threads for i in range(threads + 1)]
    st = env.state.reshape(T, -1)

    def run(i):
        lo, hi = cuts[i], cuts[i + 1]
        sl, ep = C.c_int64(0), C.c_int64(0)
        part = st[lo:hi]
        plies = lib().ddzo_rollout_random(_ptr(part), hi - lo, env.seed, env.gid_base + lo, int(n_iters),
                                          C.byref(sl), C.byref(ep))
        return plies, sl.value, ep.value

    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(run, range(threads)))
    return tuple(sum(r[k] for r in res) for k in range(3))


def rows_to_onehot(rows):
    r = np.ascontiguousarray(rows, np.int8).reshape(-1, ROW)
    out = np.zeros((r.shape[0], 15, 4), np.float32)
    lib().ddzo_rows_to_onehot(_ptr(r), r.shape[0], _ptr(out))
    return out


# ---- rule-based opponent (oracle/ddz_auto_oracle.c) ----
def cards_value():
    """cards_value of rule_based/utils/evaluator.py:10-47 as float64[13527]."""
    v = np.zeros(NUM_ACTIONS, np.int16)
    lib().ddzo_cards_value_x2(_ptr(v))
    return v.astype(np.float64) / 2.0


def _unpack_combs(buf, used):
    out, p = [], 0
    while p < used:
        n = int(buf[p])
        out.append(buf[p + 1:p + 1 + n].tolist())
        p += 1 + n
    return out


def _combinations(fn, mask, target, cap=1 << 22):
    mask = np.ascontiguousarray(mask, np.uint8)
    target = np.ascontiguousarray(target, np.uint8)
    buf = np.zeros(cap, np.int32)
    nc = C.c_int64(0)
    used = fn(_ptr(mask), mask.shape[0], _ptr(target), _ptr(buf), cap, C.byref(nc))
    if used > cap:
        buf = np.zeros(used, np.int32)
        used = fn(_ptr(mask), mask.shape[0], _ptr(target), _ptr(buf), used, C.byref(nc))
    return _unpack_combs(buf, used)


def combinations_recursive(mask, target):
    """stand-in for the absent native env.get_combinations_recursive (decomposer spec v1)."""
    return _combinations(lib().ddzo_combinations_recursive, mask, target)


def combinations_nosplit(mask, card_mask):
    """stand-in for the absent native env.get_combinations_nosplit (decomposer spec v1)."""
    return _combinations(lib().ddzo_combinations_nosplit, mask, card_mask)


def auto_combinations(hand15, follow):
    h = np.ascontiguousarray(hand15, np.int8)
    cap = 1 << 22
    buf = np.zeros(cap, np.int32)
    nc = C.c_int64(0)
    used = lib().ddzo_auto_combinations(_ptr(h), int(bool(follow)), _ptr(buf), cap, C.byref(nc))
    assert used <= cap
    return _unpack_combs(buf, used)


def auto_choose(hand15, last15, left3, role, want_stats=False):
    """RuleBasedModel.choose (rule_based_model.py:43-101) -> canonical action id."""
    h = np.ascontiguousarray(hand15, np.int8)
    l = None if last15 is None else np.ascontiguousarray(last15, np.int8)
    lf = np.ascontiguousarray(left3, np.int32)
    st = np.zeros(2, np.int64)
    a = lib().ddzo_auto_choose(_ptr(h), _ptr(l), _ptr(lf), int(role), _ptr(st))
    if a < 0:
        raise ValueError("last is not a combo of the action space")
    return (a, st) if want_stats else a


def state_prob(known60, n1, n2):
    """get_state_prob_manual (server/core.py:26-33), prob planes spec v1: f32 [2,15,4]."""
    k = np.ascontiguousarray(np.asarray(known60).reshape(60) != 0, np.uint8)
    out = np.zeros((2, 15, 4), np.float32)
    lib().ddzo_state_prob(_ptr(k), int(n1), int(n2), _ptr(out))
    return out
