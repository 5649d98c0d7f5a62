"""ctypes binding of libddz_hip.so (C ABI: include/ddz_env.h).

There is NO CPU fallback: if the library is missing or no MI355X is visible, every
product entry point raises.  (The CPU oracle under oracle/ is test infrastructure and is
never imported from here.)"""
import ctypes as C
import os

from .build import LIB as _DEFAULT_LIB, LIB_JK as _JK_LIB

_libs = {}

SYMBOLS = {
    # name: (restype, argtypes)
    "ddz_abi_version": (C.c_int, []),
    "ddz_num_actions": (C.c_int, []),
    "ddz_strerror": (C.c_char_p, [C.c_int]),
    "ddz_last_hip_error": (C.c_int, []),
    "ddz_state_bytes": (C.c_int64, [C.c_int64]),
    "ddz_scratch_bytes": (C.c_int64, [C.c_int64]),
    "ddz_face_planes": (C.c_int, [C.c_int]),
    "ddz_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64, C.c_uint64, C.c_uint64, C.c_int,
                             C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "ddz_destroy": (C.c_int, [C.c_void_p]),
    "ddz_invalidate": (C.c_int, [C.c_void_p]),
    "ddz_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_legal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ddz_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_mask_words": (C.c_int, []),
    "ddz_legal_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_legal_slab": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ddz_step_slab": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_policy_step_slab": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    "ddz_slab_to_csr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_void_p]),
    "ddz_observe": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ddz_state_prob": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ddz_rows_to_onehot": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ddz_observe_actions": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ddz_get_moves": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "ddz_get_moves_slab": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_void_p, C.c_void_p]),
    "ddz_rollout_random": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_rollout_random_csr": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_void_p]),
    "ddz_rollout_csr_staging_bytes": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "ddz_rollout_random_csr_staged": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ddz_rollout_random_timed": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int64, C.POINTER(C.c_double), C.c_void_p]),
    "ddz_read_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_select": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    "ddz_select_slab": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]),
    "ddz_q_features": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.c_void_p]),
    "ddz_q_slab": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_int64, C.c_void_p, C.c_void_p]),
    "ddz_q_features_packed": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "ddz_q_slab_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ddz_q_fc1_tile_rows": (C.c_int, []),
    "ddz_q_need_scratch_bytes": (C.c_int64, [C.c_int64]),
    "ddz_q_need": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p]),
    "ddz_q_features_needed": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ddz_q_fc1_dense": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_q_fc1_rows": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.c_void_p]),
    "ddz_q_slab_needed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ddz_q_shared_ws_bytes": (C.c_int64, []),
    "ddz_q_shared_need_ws_bytes": (C.c_int64, [C.c_int64]),
    "ddz_q_shared_need": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_q_features_drows": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ddz_q_shared_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_q_features_rows": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_q_gather_h0": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_q_fc1_rows_k": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "ddz_action_table": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "ddz_pack_trajectory": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ddz_auto_choose_state": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_auto_choose": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "ddz_debug_cards_value": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "ddz_debug_auto_leaf": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_debug_auto_choose_state": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_debug_set_geometry": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "ddz_debug_set_auto_teams": (C.c_int, [C.c_void_p, C.c_int]),
    "ddz_status": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddz_sync": (C.c_int, [C.c_int, C.c_void_p]),
    "ddz_device_status": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "ddz_debug_classify": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
}


class DdzError(RuntimeError):
    pass


_override = None


def use_library(path):
    """Diagnostics (tools/): load this build of the default library instead of csrc/libddz_hip.so -- an explicit call
    before the first use, never an environment variable."""
    global _override
    if False in _libs:
        raise DdzError("use_library() must be called before the library is first used")
    _override = path


def lib(jk=False):
    """Load libddz_hip.so (jk=True: libddz_hip_jk.so, the rule set with the 24 joker-kicker rows);
    raises (never falls back) when it is absent."""
    jk = bool(jk)
    if jk not in _libs:
        LIB = _JK_LIB if jk else (_override or _DEFAULT_LIB)
        if not os.path.exists(LIB):
            raise DdzError(
                f"{LIB} is missing: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError = header/library mismatch, loud
            fn.restype = res
            fn.argtypes = args
        if L.ddz_abi_version() != 1:
            raise DdzError("libddz_hip.so ABI version mismatch")
        if L.ddz_num_actions() != (13551 if jk else 13527):
            raise DdzError(f"{LIB}: unexpected rule set ({L.ddz_num_actions()} actions)")
        _libs[jk] = L
    return _libs[jk]


def check(rc, L=None):
    if rc != 0:
        L = L or lib()
        msg = L.ddz_strerror(rc).decode()
        if rc == -3:
            msg += f" (hipError {L.ddz_last_hip_error()})"
        raise DdzError(f"libddz_hip: {msg}")
