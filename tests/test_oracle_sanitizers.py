"""CPU: the oracle (the checker everything else is compared with) under ASan + UBSan.
GPU sanitizers are not available on the pool, so the sanitizer run is on the CPU build only."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle
oracle._libs.clear()
oracle.build = lambda force=False, jk=False: sys.argv[2]   # load the sanitizer build instead
rows, info = oracle.action_table()
g = np.load(sys.argv[1] + "/tests/golden/legal_cases.npz")
for k in range(0, len(g["hands"]), 7):
    lid = int(g["last_ids"][k])
    ids = oracle.legal(g["hands"][k], None if lid == 0 else rows[lid, :15])
    assert np.array_equal(ids, g["ids"][g["offsets"][k]:g["offsets"][k + 1]].astype(np.int32))
env = oracle.OracleEnv(97, seed=5, gid_base=2**33)
env.reset()
for it in range(90):
    env.legal()
    env.step(oracle.STEP_RANDOM, auto_reset=(it % 2 == 0), want_traj=True)
    for v in range(4):
        env.observe(v)
m = np.zeros(97, np.uint8); m[::3] = 1
env.reset(m)
env.legal(); env.select(np.zeros(env.total, np.float32), 0.5)
env.rollout_random(5)
print("sanitized-ok")
"""


def test_oracle_under_asan_ubsan(tmp_path):
    so = os.path.join(REPO, "oracle", "libddz_oracle_asan.so")
    r = subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "libddz_oracle_asan.so"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-200:])
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, str(drv), REPO, so], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "sanitized-ok" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-2000:]
