"""Timing experiment: tables per wave (= tables per block / 16) of the row stage k_q_slab_needed, in the configs[2] loop at
65,536 tables (a build with -DDDZ_QS_TPW_ENV reads DDZ_QS_TPW at launch).  python tools/row_stage_probe.py [tpw ...]"""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "build_variants")
lib = os.path.join(out, "qs_tpw_env.so")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        import torch
        importlib.import_module("doudizhu-rl_amd._lib").use_library(lib)
        pkg = importlib.import_module("doudizhu-rl_amd")
        glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
        torch.manual_seed(0)
        net = glue.QNet(6).to("cuda:0").eval()
        env = pkg.BatchedEnv(65536, seed=0)
        env.reset()
        loop = glue.PolicyLoop(env, net, face_variant=3, epsilon=0.0)
        loop.run(150)                      # the steady-state mix
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); loop.run(50); e1.record(); torch.cuda.synchronize()
        st = loop.profile(10)
        print(f"DDZ_QS_TPW={os.environ.get('DDZ_QS_TPW', 'default')}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per iteration, row stage "
              f"{st['row_stage']['us']:6.1f} us, status {env.status()}", flush=True)
    else:
        os.makedirs(out, exist_ok=True)
        if not os.path.exists(lib):
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DDDZ_QS_TPW_ENV=1", "-o", lib,
                                   os.path.join(ROOT, "doudizhu-rl_amd", "csrc", "ddz_engine.hip")])
        for v in (sys.argv[1:] or ["1", "2", "4", "8"]):
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child"], env={**os.environ, "DDZ_QS_TPW": v})
