#!/usr/bin/env python3
"""BASELINE.json configs[3]: tables vs the rule-based opponent, full episode returns.

The lord is played by a policy (default: the uniform-random policy of envi.py:79-85; --lord net: greedy arg-max of a
randomly initialised NetCooperationSimplify, net.py:137-150), both farmers by the rule agent -- what game.py:106
`self.env.step_auto()` does for every role without a network (rule_based/utils/rule_based_model.py:43-101 on the
device, DESIGN.md 4 "decomposer spec v1").  One lock-step iteration = auto_choose (k_auto) + step_slab(DDZ_STEP_IDS),
finished tables are re-dealt in the same launch.  Reports env steps/s and the mean episode return per role with the
reference's reward_dict (game.py:13-14: lord 100, farmers 50; winners +, losers -).

  python examples/config4_rule_opponent.py [--tables 65536] [--iters 200] [--lord random|net]
"""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--lord", choices=("random", "net"), default="random")
    a = ap.parse_args()
    pkg = importlib.import_module("doudizhu-rl_amd")
    dev = torch.device("cuda:0")
    T = a.tables
    env = pkg.BatchedEnv(T, seed=0, device=dev)
    env.reset()
    env.legal_slab()
    net = None
    if a.lord == "net":
        glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
        torch.manual_seed(0)
        net = glue.QNet(6).to(dev).eval()
        fq = glue.FactorisedQ(net)
        face = torch.empty((T, 6, 15, 4), dtype=torch.float32, device=dev)
        qbuf = torch.zeros((T, env.slab_stride), dtype=torch.float32, device=dev)
    stats = torch.zeros((T, 2), dtype=torch.int64, device=dev)
    nodes = torch.zeros(2, dtype=torch.int64, device=dev)

    @torch.no_grad()
    def lord_ids():
        """the Q-network's greedy move as a canonical action id, for every table (only the lord's tables use it): the
        ragged forward of dqn_glue (per-rank GEMMs over the rows the actors' hands allow + ddz_q_slab_packed), arg-max by
        ddz_select_slab"""
        env.observe(3, out=face)
        q = fq.q_slab(env, fq.needed(env, face, shared="all"), out=qbuf)   # (shared rows: csrc/ddz_qnet.h sections 5-6)
        choice = env.select_slab(q)
        return env.slab_ids().gather(1, choice.clamp(min=0).long()[:, None])[:, 0].to(torch.int32)

    def iteration(collect):
        sel = env.auto_choose(0b101, stats=stats if collect else None)
        if net is not None:
            sel = torch.where(sel >= 0, sel, lord_ids())
        env.step_slab(sel, pkg.STEP_IDS, auto_reset=True)
        if collect:
            nodes.add_(stats.sum(0))

    for _ in range(10):
        iteration(False)
    torch.cuda.synchronize()
    s0 = env.stats()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        iteration(False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = env.stats()
    d = {k: s1[k] - s0[k] for k in s1}
    for _ in range(10):  # search statistics, outside the timed region
        iteration(True)
    nd = nodes.tolist()
    eps = max(1, d["episodes"])
    lord_w, farm_w = d["lord_wins"], d["up_wins"] + d["down_wins"]
    ret = {"lord": 100.0 * (lord_w - farm_w) / eps, "up": 50.0 * (farm_w - lord_w) / eps, "down": 50.0 * (farm_w - lord_w) / eps}
    auto_plies = 10 * T * 2 / 3
    print(f"tables={T} iters={a.iters} lord={a.lord}: {dt / a.iters * 1e3:.2f} ms/iteration, "
          f"{d['plies'] / dt / 1e6:.2f} M env steps/s, episodes={d['episodes']} "
          f"(lord wins {lord_w}, farmer wins {farm_w} = up {d['up_wins']} + down {d['down_wins']}), "
          f"mean episode return lord {ret['lord']:+.1f} up {ret['up']:+.1f} down {ret['down']:+.1f}, "
          f"rule agent: {nd[0] / auto_plies:.0f} combinations / {nd[1] / auto_plies:.0f} search nodes per decision, "
          f"status={env.status()}")


if __name__ == "__main__":
    main()
