#!/usr/bin/env python3
"""Batched DQN training of the landlord against random farmers: the reference's
loop (game.py:90-181, dqn.py:21-71) for T tables at once on one MI355X.

Per lock-step iteration
  legal() -> observe(planes of EnvCooperationSimplify) -> Q(face, action) over the ragged legal list of
  the tables where the landlord is to move (the farmers' rows get uniform random values, so the same
  segment-argmax plays them at random, envi.py:79-85) -> ddz_select (epsilon-greedy, dqn.py:50-71)
  -> TransitionAssembler.before_step (closes the landlord's previous transition with s1 = face now,
  a1 = greedy action now, game.py:109-127) -> step(auto_reset=False) -> after_step at terminal plies
  (+/-reward_dict, a1 = zeros, game.py:113-123) -> reset(done) -> replay sample -> TD step
  (dqn.py:33-48: y = r + gamma * Q_target(s1, a1), MSE, Adam lr 1e-4) -> target sync.
Hyper-parameters are the reference's (config.py:8-14).  Random init, no dataset: this shows the hot
path feeding a learner, it is not a benchmark of the network.

  python examples/config3_dqn_train.py [--tables 4096] [--iters 200]
"""
import argparse
import importlib
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

GAMMA, EPS_HIGH, EPS_LOW, DECAY = 0.95, 0.5, 0.01, int((8000 * (2 / 3)) / 5)  # config.py:8-13
REPLAY_SIZE, BATCH_SIZE, UPDATE_TARGET_EVERY = 20000, 256, 20                   # config.py:11-14


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    pkg = importlib.import_module("doudizhu-rl_amd")
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    dev = torch.device("cuda:0")
    torch.manual_seed(a.seed)
    policy, target = glue.QNet(6).to(dev), glue.QNet(6).to(dev)
    target.load_state_dict(policy.state_dict())
    target.eval()
    opt = torch.optim.Adam(policy.parameters(), lr=1e-4)  # dqn.py:19
    T, P = a.tables, 6
    env = pkg.BatchedEnv(T, seed=a.seed, device=dev)
    env.reset()
    asm = glue.TransitionAssembler(T, P, dev)
    replay = glue.Replay(REPLAY_SIZE, P, dev)
    tables = torch.arange(T, device=dev)
    losses, t0, episodes, lord_wins = [], time.perf_counter(), 0, 0
    for it in range(a.iters):
        eps = EPS_LOW + (EPS_HIGH - EPS_LOW) * math.exp(-episodes / T / DECAY)  # dqn.py:73-75 per table-episode
        offsets, rows, _ = env.legal()
        face = env.observe(3)
        role = env.role.long()
        lord = role == 1
        total = int(offsets[-1].item())                        # (a training loop syncs anyway: replay, logging)
        seg = torch.repeat_interleave(tables, offsets.diff().long(), output_size=total)
        acts = pkg.rows_to_onehot(rows[:total])
        policy.eval()
        qnet = glue.ragged_q(policy, face, rows[:total], offsets)   # Q of every legal row, first layer factorised
        q = torch.where(lord[seg], qnet, torch.rand(total, device=dev))   # farmers: random play
        greedy_idx = env.select(q, 0.0)
        choice = env.select(q, eps)
        first = offsets[:-1].long()
        chosen, greedy = acts[first + choice.long()], acts[first + greedy_idx.long()]
        replay.push(asm.before_step(role, face, chosen, greedy, active=lord))
        done, r, _ = env.step(choice, pkg.STEP_CHOICE, auto_reset=False)
        if bool(done.any()):
            replay.push(asm.after_step(role, done, r, env.observe(3)))
            episodes += int(done.sum())
            lord_wins += int((done.bool() & (r < 0)).sum())
            env.reset(mask=done)
        if replay.n >= BATCH_SIZE:
            policy.train()                                      # dqn.py:44: the policy net trains with dropout on
            losses.append(float(glue.td_step(policy, target, opt, replay.sample(BATCH_SIZE), GAMMA)))
        if (it + 1) % UPDATE_TARGET_EVERY == 0:
            target.load_state_dict(policy.state_dict())      # dqn.py:77-79
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"tables": T, "iters": a.iters, "ms_per_iteration": dt / a.iters * 1e3, "episodes": episodes,
           "lord_win_rate": lord_wins / max(1, episodes), "replay": replay.n,
           "first_loss": losses[0] if losses else None, "last_loss": losses[-1] if losses else None,
           "status": env.status()}
    print(out)
    return out


if __name__ == "__main__":
    main()
