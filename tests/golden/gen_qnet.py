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
