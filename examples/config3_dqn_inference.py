#!/usr/bin/env python3
"""BASELINE.json configs[2]: tables on one MI355X with Q-net inference in the loop (SURVEY 8d "Config 3").

Per lock-step iteration (doudizhu-rl_amd/dqn_glue.py PolicyLoop, nothing on the host in between):
  face (EnvCooperationSimplify planes) -> FactorisedQ.tables: the first layer per (table, rank, count) as dense GEMMs
  -> ddz_q_slab: Q of EVERY legal action of every table over the slab lists -> ddz_policy_step_slab: greedy arg-max,
  apply, next lists, next face in one launch.
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
    a = ap.parse_args(argv)
    pkg = importlib.import_module("doudizhu-rl_amd")
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = glue.QNet(6).to(dev).eval()
    T = a.tables
    env = pkg.BatchedEnv(T, seed=0, device=dev)
    env.reset()
    loop = glue.PolicyLoop(env, net, face_variant=3, epsilon=0.0)
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
    return out


if __name__ == "__main__":
    main()
