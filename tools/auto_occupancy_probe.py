"""k_auto2 with more waves per CU: does the rule agent's launch get faster when a SIMD holds three waves instead of two?
The product geometry is LDS-bound (8 waves x 17.7 KB + the shared tables = 156 KB: one block per CU).  This probe builds
variants with smaller per-wave buffers (-DDDZ_STAGE_CAP / -DDDZ_A2_CAP / -DDDZ_A2_BOX: NOT safe for every hand -- a timing
experiment on the states of the configs[3] loop, whose status word it prints) and more waves per block, and times
auto_choose on mid-game states of 65,536 tables.
  python tools/auto_occupancy_probe.py [T=65536]"""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "doudizhu-rl_amd", "csrc")
out = os.path.join(ROOT, "build_variants")
os.makedirs(out, exist_ok=True)
VARIANTS = {
    "product": [],
    "cap104_box32": ["-DDDZ_A2_CAP=104", "-DDDZ_A2_BOX=32"],
    "wpb7_cap128": ["-DDDZ_A2_CAP=128", "-DDDZ_A2_WPB=7"],
    "wpb8_small": ["-DDDZ_STAGE_CAP=320", "-DDDZ_A2_CAP=64", "-DDDZ_A2_BOX=32"],
    "wpb10": ["-DDDZ_STAGE_CAP=384", "-DDDZ_A2_CAP=64", "-DDDZ_A2_BOX=32", "-DDDZ_A2_WPB=10", "-DDDZ_A2_OCC=2"],
    "wpb11": ["-DDDZ_STAGE_CAP=320", "-DDDZ_A2_CAP=64", "-DDDZ_A2_BOX=32", "-DDDZ_A2_WPB=11", "-DDDZ_A2_OCC=3"],
    "wpb12": ["-DDDZ_STAGE_CAP=208", "-DDDZ_A2_CAP=64", "-DDDZ_A2_BOX=32", "-DDDZ_A2_WPB=12", "-DDDZ_A2_OCC=3"],
}


def build(name):
    lib = os.path.join(out, f"auto_{name}.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", *VARIANTS[name],
                           "-o", lib, os.path.join(csrc, "ddz_engine.hip")])
    return lib


if __name__ == "__main__":
    if len(sys.argv) > 2:    # child: one variant
        name, T = sys.argv[1], int(sys.argv[2])
        import torch
        importlib.import_module("doudizhu-rl_amd._lib").use_library(os.path.join(out, f"auto_{name}.so"))
        pkg = importlib.import_module("doudizhu-rl_amd")
        env = pkg.BatchedEnv(T, seed=0)
        env.reset(); env.legal_slab()
        for it in range(12):
            env.step_auto(0b101, slab=True)
        torch.cuda.synchronize()
        snap = env.state_export().clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        res = []
        for roles in (0b101, 0b111):
            best = 1e9
            for _ in range(5):
                e0.record(); ids = env.auto_choose(roles); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            res.append((roles, best, int(ids.to(torch.int64).clamp(min=0).sum())))
        # the loop itself: 40 iterations
        torch.cuda.synchronize()
        e0.record()
        for it in range(40):
            env.step_auto(0b101, slab=True)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:12s} T={T}: auto_choose(farmers) {res[0][1] * 1e3:7.0f} us, (all roles) {res[1][1] * 1e3:7.0f} us; "
              f"loop {e0.elapsed_time(e1) / 40 * 1e3:7.0f} us per iteration = {T * 40 / e0.elapsed_time(e1) / 1e3:6.1f} M steps/s; "
              f"id checksums {res[0][2]} {res[1][2]}, status {env.status()}")
    else:
        T = sys.argv[1] if len(sys.argv) > 1 else "65536"
        for name in VARIANTS:
            try:
                build(name)
            except subprocess.CalledProcessError:
                print(f"{name}: does not build (LDS?)")
                continue
            subprocess.call([sys.executable, os.path.abspath(__file__), name, T])
