#!/usr/bin/env python3
"""Which geometry of k_fc1 (csrc/ddz_qnet.h) is fastest: builds the library once per variant (-DDDZ_FC_*) into
build_variants/ and times ddz_q_fc1_dense on the configs[2] shape ([65536, 3840] x [3840, 256], fp32) in a child process per
variant, beside torch.addmm (hipBLASLt) on the same operands.
  python tools/fc1_probe.py            (on the GPU box: builds, then measures)
  python tools/fc1_probe.py --build    (build only, e.g. in the build container before gpurun ships the tree)"""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {  # name: (waves, tm, tn, kc, occ)
    "w4_1x8_k16_occ2": (4, 1, 8, 16, 2),      # round 4's first version: 128-row tile, 32 x 256 per wave, two blocks per CU
    "w4_4x4_k16_occ1": (4, 4, 4, 16, 1),      # 256-row tile, 128 x 128 per wave (256 accumulator registers), one block per CU
    "w8_2x4_k16_occ1": (8, 2, 4, 16, 1),      # 256-row tile, eight waves of 64 x 128
    "w8_2x4_k32_occ1": (8, 2, 4, 32, 1),
    "w4_2x4_k16_occ2": (4, 2, 4, 16, 2),      # 128-row tile, 64 x 128 per wave, two blocks per CU
    "w8_1x8_k16_occ1": (8, 1, 8, 16, 1),      # 256-row tile, eight waves of 32 x 256
}
OUT = os.path.join(ROOT, "build_variants")


def build():
    b = importlib.import_module("doudizhu-rl_amd.build")
    os.makedirs(OUT, exist_ok=True)
    jobs = []
    for name, (w, tm, tn, kc, occ) in VARIANTS.items():
        lib = os.path.join(OUT, f"libddz_hip_fc1_{name}.so")
        extra = [f"-DDDZ_FC_WAVES={w}", f"-DDDZ_FC_TM={tm}", f"-DDDZ_FC_TN={tn}", f"-DDDZ_FC_KC={kc}", f"-DDDZ_FC_OCC={occ}"]
        if not os.path.exists(lib) or b.stale(lib):
            jobs.append((name, subprocess.Popen(b._cmd(lib, extra), cwd=b.CSRC)))
        if len(jobs) >= 6:
            for n, p in jobs:
                assert p.wait() == 0, n
            jobs = []
    for n, p in jobs:
        assert p.wait() == 0, n


CHILD = r'''
import importlib, sys, time
sys.path.insert(0, %r)
L = importlib.import_module("doudizhu-rl_amd._lib")
L.use_library(%r)
import torch
pkg = importlib.import_module("doudizhu-rl_amd")
T, K = 65536, 3840
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn((T, K), device="cuda", generator=g)
w = torch.randn((K, 256), device="cuda", generator=g)
c = torch.zeros((T, 256), device="cuda")
ref = (a[:512].double() @ w.double())
pkg.q_fc1_dense(a, w, c)
err = float((c[:512].double() - ref).abs().max() / ref.abs().max())
for _ in range(3): pkg.q_fc1_dense(a, w, c)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): pkg.q_fc1_dense(a, w, c)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print("%%-18s %%8.1f us  %%6.1f TFLOP/s  tile %%d rows  rel err %%.1e" %% (%r, us, 2.0 * T * K * 256 / us / 1e6, L.lib().ddz_q_fc1_tile_rows(), err))
if %r:
    c.zero_()
    for _ in range(3): c.addmm_(a, w)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): c.addmm_(a, w)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("%%-18s %%8.1f us  %%6.1f TFLOP/s" %% ("torch.addmm", us, 2.0 * T * K * 256 / us / 1e6))
'''


def main():
    build()
    if "--build" in sys.argv:
        return
    first = True
    for name in VARIANTS:
        lib = os.path.join(OUT, f"libddz_hip_fc1_{name}.so")
        subprocess.run([sys.executable, "-c", CHILD % (ROOT, lib, name, first)], check=False)
        first = False


if __name__ == "__main__":
    main()
