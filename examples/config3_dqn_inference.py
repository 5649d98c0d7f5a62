#!/usr/bin/env python3
"""BASELINE.json configs[2]: tables on one MI355X with Q-net inference in the loop (SURVEY 8d "Config 3").

Per lock-step iteration (doudizhu-rl_amd/dqn_glue.py PolicyLoop, needed-rows form, nothing on the host in between):
  face (EnvCooperationSimplify planes) -> ddz_q_need (the (rank, count) rows the legal moves use) -> ddz_q_features_needed
  (first layer) -> ddz_q_fc1_dense + ddz_q_fc1_rows (fc1 on the fp32 MFMA kernel k_fc1) -> ddz_q_slab_needed: Q of EVERY
  legal action of every table over the slab lists -> ddz_policy_step_slab: greedy arg-max, apply, next lists, next face.
The network has the architecture and parameter names of the reference's NetCooperationSimplify (net.py:137-150, forward
net.py:81-102), randomly initialised (no trained weights ship with the reference), eval mode.

  python examples/config3_dqn_inference.py [--tables 65536] [--iters 20]
"""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--mode", default="needed", choices=("needed", "packed", "full"))
    ap.add_argument("--gemm", default="torch", choices=("mfma", "torch"), help="the plain dense GEMM: hipBLASLt (default) or the engine's k_fc1")
    ap.add_argument("--dense", action="store_true", help="H0 by the dense K = 3840 GEMM over every table instead of the shared rows")
    ap.add_argument("--stages", action="store_true", help="also print the per-stage device times (HIP events)")
    a = ap.parse_args(argv)
    pkg = importlib.import_module("doudizhu-rl_amd")
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = glue.QNet(6).to(dev).eval()
    T = a.tables
    env = pkg.BatchedEnv(T, seed=0, device=dev)
    env.reset()
    loop = glue.PolicyLoop(env, net, face_variant=3, epsilon=0.0, mode=a.mode, gemm=a.gemm,
                           shared=False if (a.dense or a.mode != "needed") else None)
    loop.run(2)
    torch.cuda.synchronize()
    s0 = env.stats()
    t0 = time.perf_counter()
    loop.run(a.iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = env.stats()
    rows = s1["legal_rows"] - s0["legal_rows"]
    out = {"tables": T, "iters": a.iters, "ms_per_iteration": dt / a.iters * 1e3, "env_steps_per_s": T * a.iters / dt,
           "legal_rows_per_table": rows / a.iters / T, "q_evals_per_s": rows / dt, "episodes": s1["episodes"],
           "status": env.status()}
    print(out)
    if a.stages and a.mode == "needed":
        for k, v in loop.profile(10).items():
            rate = f"{v['flop'] / v['us'] / 1e6:8.1f} TFLOP/s" if v.get("flop") else f"{v['bytes'] / v['us'] / 1e3:8.1f} GB/s"
            print(f"  {k:12s} {v['us']:9.1f} us  {rate}  {v['kernel']}")
    return out


if __name__ == "__main__":
    main()
