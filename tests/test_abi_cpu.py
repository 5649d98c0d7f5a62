"""CPU: the C-ABI library loads and exports every symbol include/ddz_env.h declares; host
logic that needs no GPU (sizes, error paths, codecs of the envi.py mirror)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import importlib
    build = importlib.import_module("doudizhu-rl_amd.build")
    build.build()
    return importlib.import_module("doudizhu-rl_amd")


def test_header_symbols_exported(pkg):
    import importlib
    lib = importlib.import_module("doudizhu-rl_amd._lib")
    hdr = open(os.path.join(REPO, "include", "ddz_env.h")).read()
    declared = set(re.findall(r"\b(ddz_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 18
    L = lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    assert L.ddz_abi_version() == 1


def test_sizes_and_error_paths(pkg):
    import importlib
    L = importlib.import_module("doudizhu-rl_amd._lib").lib()
    assert L.ddz_state_bytes(4096) == 4096 * 11 * 16
    assert L.ddz_scratch_bytes(4096) >= 4096 * 24 and L.ddz_scratch_bytes(4096) % 256 == 0
    assert [L.ddz_face_planes(v) for v in range(4)] == [4, 7, 9, 6]
    assert L.ddz_face_planes(4) == -1
    assert L.ddz_strerror(-2) == b"bad handle"
    h = C.c_void_p()
    assert L.ddz_create(C.byref(h), 0, 0, 0, 0, None, 0, None, 0) == -1      # EINVAL
    assert L.ddz_destroy(None) == -2                                          # EHANDLE
    assert L.ddz_legal(None, None, None, None, 0, None) == -2
    assert L.ddz_rows_to_onehot(0, None, 5, None, None) == -1


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.DdzError):
        pkg.BatchedEnv(4)
    with pytest.raises(pkg.DdzError):
        pkg.rows_to_onehot(torch.zeros((1, 16), dtype=torch.int8))


def test_product_does_not_import_oracle():
    pkgdir = os.path.join(REPO, "doudizhu-rl_amd")
    for root, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "ddz_oracle" not in src and "import oracle" not in src, f
                assert "from oracle" not in src, f


def test_codecs_match_envi_semantics(pkg):
    E = pkg.Env
    arr = np.array([0, 2, 0, 0, 1, 0, 0, 0, 0, 0, 0, 3, 0, 1, 1])
    cards = E.arr2cards(arr)                       # envi.py:118-130
    assert cards.tolist() == [4, 4, 7, 14, 14, 14, 16, 17]
    assert np.array_equal(E.cards2arr(cards), arr)  # envi.py:132-137
    oh = E.batch_arr2onehot([arr, np.zeros(15)])    # envi.py:139-146
    assert oh.shape == (2, 15, 4) and oh[0, 11].tolist() == [1, 1, 1, 0] and oh[1].sum() == 0
    assert np.array_equal(E.onehot2arr(oh[0]), arr)  # envi.py:148-157


def test_thermometer_codec_matches_golden(pkg, golden):
    bits = np.unpackbits(golden("thermo.npz")["bits"], axis=1)[:, :60]
    rows = golden("action_table.npz")["rows"]
    oh = pkg.Env.batch_arr2onehot(rows[::53]).reshape(-1, 60)
    assert np.array_equal(oh, bits[::53])
