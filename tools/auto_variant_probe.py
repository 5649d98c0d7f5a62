"""A/B of k_auto2 builds that differ by macros (-D...): the rule agent's decisions on mid-game states of the configs[3] loop
(farmers = rule agent, lord = engine RNG) -- auto_choose alone, the loop, and id checksums (results never depend on a variant).
Libraries are built into build_variants/ when missing (build them in the build container: they travel with gpurun).
  python tools/auto_variant_probe.py name[:-Dflag[,-Dflag...]] ...      e.g.  product scan2:-DA2_SCAN_ROUNDS=2"""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "doudizhu-rl_amd", "csrc")
out = os.path.join(ROOT, "build_variants")


def lib_of(name):
    return os.path.join(out, f"autov_{name}.so")


def build(name, flags):
    os.makedirs(out, exist_ok=True)
    if not os.path.exists(lib_of(name)):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", *flags,
                               "-o", lib_of(name), os.path.join(csrc, "ddz_engine.hip")])


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        name, T = sys.argv[2], int(sys.argv[3])
        import torch
        importlib.import_module("doudizhu-rl_amd._lib").use_library(lib_of(name))
        pkg = importlib.import_module("doudizhu-rl_amd")
        env = pkg.BatchedEnv(T, seed=0)
        env.reset(); env.legal_slab()
        for it in range(12):
            env.step_auto(0b101, slab=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        res = []
        for roles in (0b101, 0b111):
            ts = []
            for _ in range(7):
                e0.record(); ids = env.auto_choose(roles); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res.append((min(ts), sorted(ts)[3], int(ids.to(torch.int64).clamp(min=0).sum())))
        e0.record()
        for it in range(60):
            env.step_auto(0b101, slab=True)
        e1.record(); torch.cuda.synchronize()
        h = int(env.state_export().to(torch.int64).sum())
        print(f"{name:12s} T={T}: auto_choose(farmers) {res[0][0] * 1e3:6.0f} us (median {res[0][1] * 1e3:6.0f}), all roles {res[1][0] * 1e3:6.0f} us; "
              f"loop {e0.elapsed_time(e1) / 60 * 1e3:6.0f} us per iteration = {T * 60 / e0.elapsed_time(e1) / 1e3:6.1f} M steps/s; "
              f"checksums {res[0][2]} {res[1][2]} {h}, status {env.status()}", flush=True)
    else:
        specs = [a for a in sys.argv[1:] if not a.startswith("--")]
        T = next((a[4:] for a in sys.argv[1:] if a.startswith("--T=")), "65536")
        only_build = "--build-only" in sys.argv
        for spec in specs:
            name, _, fl = spec.partition(":")
            build(name, [f for f in fl.split(",") if f])
            if not only_build:
                subprocess.call([sys.executable, os.path.abspath(__file__), "--child", name, T])
