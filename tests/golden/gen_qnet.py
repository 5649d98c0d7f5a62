#!/usr/bin/env python3
"""Generate tests/golden/qnet.npz (fixture G9) from the reference's OWN networks and TD arithmetic.  Runs ONLY in the
build container: it imports /root/reference/net.py (which imports config.py; both execute definitions only) and writes
plain data -- inputs and the outputs the reference computes for them -- never reference source.
    python tests/golden/gen_qnet.py

For each of the four Q-networks of the reference (net.py:66-150: NetComplicated 5 input planes, NetMoreComplicated 8,
NetCooperation 10, NetCooperationSimplify 7), built right after torch.manual_seed(seed) and put in eval mode:
  face [n,P,15,4], actions [n,15,4] (thermometers / fractions as envi.py produces them), q = net(face, actions) [n],
  q_single = net(face[0], actions[:m]) -- the calling convention of dqn.py:56 (one state, all its actions) --
  and a checksum of every parameter (sum, sum of squares), so that a test can tell "the same initial weights" apart from
  "a different network".
Plus the TD target of dqn.py:40-41 for a few (r, done, q_next) triples with conf.GAMMA.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True  # never write into /root/reference
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
import config as rconf  # noqa: E402  (reference, container only)
import net as rnet      # noqa: E402

CLASSES = {4: rnet.NetComplicated, 7: rnet.NetMoreComplicated, 9: rnet.NetCooperation, 6: rnet.NetCooperationSimplify}
out = {}
for P, cls in CLASSES.items():
    seed = 1000 + P
    torch.manual_seed(seed)
    net = cls().eval()
    g = torch.Generator().manual_seed(seed + 1)
    n, m = 24, 7
    cnt = torch.randint(0, 5, (n, P, 15), generator=g)
    cnt[:, :, 13:] = cnt[:, :, 13:].clamp(max=1)
    face = (cnt[..., None] > torch.arange(4)[None, None, None, :]).float()          # thermometers (envi.py:139-146)
    face[:, -2:] *= torch.rand(n, 1, 1, 1, generator=g)                              # the two probability planes
    acnt = (torch.rand(n, 15, generator=g) < 0.25) * torch.randint(1, 5, (n, 15), generator=g)
    acnt[:, 13:] = acnt[:, 13:].clamp(max=1)
    acnt[0] = 0                                                                      # a pass
    actions = (acnt[..., None] > torch.arange(4)[None, None, :]).float()
    with torch.no_grad():
        q = net(face, actions)[:, 0]
        q_single = net(face[0], actions[:m])[:, 0]
    out[f"p{P}_seed"] = np.int64(seed)
    out[f"p{P}_face"] = face.numpy()
    out[f"p{P}_actions"] = actions.numpy()
    out[f"p{P}_q"] = q.numpy()
    out[f"p{P}_q_single"] = q_single.numpy()
    names = sorted(net.state_dict())
    out[f"p{P}_param_names"] = np.array(names)
    out[f"p{P}_param_sums"] = np.array([[float(net.state_dict()[k].double().sum()), float((net.state_dict()[k].double() ** 2).sum())]
                                         for k in names])
    out[f"p{P}_param_shapes"] = np.array([str(tuple(net.state_dict()[k].shape)) for k in names])
# dqn.py:40-41: y_true = r1 + (1 - done) * conf.GAMMA * s1_reward
r = torch.tensor([0.0, 100.0, -50.0, 0.0, 50.0])
done = torch.tensor([0.0, 1.0, 1.0, 0.0, 1.0])
qn = torch.tensor([0.3, -1.2, 7.0, -0.05, 2.5])
out["td_r"], out["td_done"], out["td_qnext"] = r.numpy(), done.numpy(), qn.numpy()
out["td_y"] = (r + (1 - done) * rconf.GAMMA * qn).numpy()
out["gamma"] = np.float64(rconf.GAMMA)
out["hyper"] = np.array([rconf.EPSILON_HIGH, rconf.EPSILON_LOW, rconf.REPLAY_SIZE, rconf.BATCH_SIZE, rconf.DECAY,
                         rconf.UPDATE_TARGET_EVERY], dtype=np.float64)
np.savez_compressed(os.path.join(HERE, "qnet.npz"), **out)
print("wrote qnet.npz:", {k: v.shape for k, v in out.items() if k.endswith("_q")}, os.path.getsize(os.path.join(HERE, "qnet.npz")), "bytes")

# ---- G10: one DQNFirst.perceive() update (dqn.py:21-48) of the reference's own agent, as numbers
import random  # noqa: E402

import dqn as rdqn  # noqa: E402  (reference, container only)

torch.manual_seed(4242)
agent = rdqn.DQNFirst(rnet.NetCooperationSimplify)     # policy + target (train mode: dropout is active, net.py / dqn.py as written)
g = torch.Generator().manual_seed(77)
B, P = rconf.BATCH_SIZE, 6


def therm(cnt):
    return (cnt[..., None] > torch.arange(4)).float()


s0 = therm(torch.randint(0, 5, (B, P, 15), generator=g)); s1 = therm(torch.randint(0, 5, (B, P, 15), generator=g))
a0 = therm((torch.rand(B, 15, generator=g) < 0.2) * torch.randint(1, 5, (B, 15), generator=g))
a1 = therm((torch.rand(B, 15, generator=g) < 0.2) * torch.randint(1, 5, (B, 15), generator=g))
rew = torch.tensor([0.0, 100.0, -100.0, 50.0, -50.0])[torch.randint(0, 5, (B,), generator=g)]
done = rew != 0
a1[done] = 0                                           # game.py:122-123: zeros at the terminal ply
for i in range(B - 1):
    agent.replay_buffer.append((s0[i], a0[i], float(rew[i]), s1[i], a1[i], bool(done[i])))
random.seed(99)
order = random.sample(range(B), B)                     # the positions perceive()'s own random.sample will draw
random.seed(99)
loss = agent.perceive(s0[B - 1], a0[B - 1], float(rew[B - 1]), s1[B - 1], a1[B - 1], bool(done[B - 1]))
sd = agent.policy_net.state_dict()
names = sorted(sd)
more = {"td_seed": np.int64(4242), "td_order": np.array(order, np.int64), "td_loss": np.float64(loss),
        "td_s0": s0.numpy().astype(np.uint8), "td_a0": a0.numpy().astype(np.uint8), "td_s1": s1.numpy().astype(np.uint8),
        "td_a1": a1.numpy().astype(np.uint8), "td_rew": rew.numpy(), "td_done_b": done.numpy(),
        "td_after_names": np.array(names),
        "td_after_sums": np.array([[float(sd[k].double().sum()), float((sd[k].double() ** 2).sum())] for k in names]),
        "td_lr": np.float64(1e-4)}
out.update(more)
# ---- the epsilon schedule (dqn.py:73-76) and the checkpoint directory convention (config.py:30-31), as the reference computes them
eps_at = [0, 1, 100, 1066, 5000, 20000]
eps = []
for e in eps_at:
    agent.update_epsilon(e)
    eps.append(float(agent.epsilon))
out["eps_episodes"], out["eps_values"] = np.array(eps_at, np.int64), np.array(eps, np.float64)
nd_names = ["0805_1409_lord_4000", "0808_0854_up_3000_61", "a_b", "plain", "x_y_z_w_v"]
out["name_dir_in"] = np.array(nd_names)
out["name_dir_out"] = np.array([rconf.name_dir(n) for n in nd_names])
out["name_dir_out_split1"] = np.array([rconf.name_dir(n, 1) for n in nd_names])
np.savez_compressed(os.path.join(HERE, "qnet.npz"), **out)
print("G10: loss", loss, "file", os.path.getsize(os.path.join(HERE, "qnet.npz")), "bytes")
