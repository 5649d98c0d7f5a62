#!/usr/bin/env python3
"""BASELINE.json configs[2]: tables on one MI355X with Q-net inference in the loop.

Per lock-step iteration: legal() (CSR list) -> observe(EnvCooperationSimplify planes) -> Q(face, action)
for every legal action of every table -> greedy choice per table (ddz_select) -> step(choice).
The network has the architecture and parameter names of the reference's NetCooperationSimplify
(net.py:137-150, forward net.py:81-102), randomly initialised (no trained weights ship with the
reference), eval mode.  Its (1,k)/stride-(1,4) convolutions over a width-4 input produce a single
column, so they are evaluated as GEMMs (rocBLAS/hipBLASLt) -- same math as nn.Conv2d, no MIOpen.

  python examples/config3_dqn_inference.py [--tables 65536] [--iters 5]
"""
import argparse
import importlib
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class QNetSimplify(nn.Module):
    """state_dict-compatible with the reference's NetCooperationSimplify (7 input planes)."""

    def __init__(self, c_in=7):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, 256, (1, 1), (1, 4))
        self.conv2 = nn.Conv2d(c_in, 256, (1, 2), (1, 4))
        self.conv3 = nn.Conv2d(c_in, 256, (1, 3), (1, 4))
        self.conv4 = nn.Conv2d(c_in, 256, (1, 4), (1, 4))
        self.conv_shunzi = nn.Conv2d(c_in, 256, (15, 1), 1)
        self.fc1 = nn.Linear(256 * (15 + 4), 256)
        self.fc2 = nn.Linear(256, 1)

    def forward_conv(self, x):
        """literal nn.Conv2d evaluation (net.py:92-101 without dropout), for the equality test"""
        n = x.shape[0]
        y = torch.cat([f(x) for f in (self.conv1, self.conv2, self.conv3, self.conv4)], -1)   # [n,256,15,4]
        y = F.max_pool2d(y, (1, 4)).reshape(n, -1)                                              # [n,3840]
        z = self.conv_shunzi(x).reshape(n, -1)                                                  # [n,1024]
        return self.fc2(F.relu(self.fc1(torch.cat([y, z], -1))))

    def forward(self, x):
        """same function as GEMMs: x [n, C, 15, 4] -> q [n, 1]"""
        n, c = x.shape[0], x.shape[1]
        rows = x.permute(0, 2, 1, 3).reshape(n * 15, c * 4)            # per rank: [C x 4] window
        w = x.new_zeros((c * 4, 4 * 256))
        for k, conv in enumerate((self.conv1, self.conv2, self.conv3, self.conv4)):
            wk = conv.weight[:, :, 0, :]                               # [256, C, k+1]
            w.view(c, 4, 4, 256)[:, : k + 1, k, :] = wk.permute(1, 2, 0)
        b = torch.cat([f.bias for f in (self.conv1, self.conv2, self.conv3, self.conv4)])
        y = (rows @ w + b).view(n, 15, 4, 256).amax(dim=2)             # max over the four convs
        y = y.permute(0, 2, 1).reshape(n, 256 * 15)                    # channel-major like the reference
        ws = self.conv_shunzi.weight[:, :, :, 0].reshape(256, c * 15)  # [256, C*15]
        cols = x.permute(0, 3, 1, 2).reshape(n * 4, c * 15)
        z = (cols @ ws.t() + self.conv_shunzi.bias).view(n, 4, 256).permute(0, 2, 1).reshape(n, 1024)
        return self.fc2(F.relu(self.fc1(torch.cat([y, z], -1))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--chunk", type=int, default=131072, help="legal rows per NN forward")
    a = ap.parse_args()
    pkg = importlib.import_module("doudizhu-rl_amd")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = QNetSimplify().to(dev).eval()
    T = a.tables
    env = pkg.BatchedEnv(T, seed=0, device=dev)
    env.reset()
    face = torch.empty((T, 6, 15, 4), dtype=torch.float32, device=dev)
    tables = torch.arange(T, device=dev)

    @torch.no_grad()
    def iteration():
        offsets, rows, _ = env.legal()
        env.observe(3, out=face)
        total = int(offsets[-1].item())                               # one host sync per iteration
        seg = torch.repeat_interleave(tables, offsets.diff().long(), output_size=total)
        q = torch.empty(total, dtype=torch.float32, device=dev)
        for lo in range(0, total, a.chunk):
            hi = min(total, lo + a.chunk)
            acts = pkg.rows_to_onehot(rows[lo:hi])                    # [n,15,4]
            x = torch.cat([face[seg[lo:hi]], acts[:, None]], dim=1)   # [n,7,15,4]  (net.py:89-90)
            q[lo:hi] = net(x)[:, 0]
        choice = env.select(q)                                        # greedy (dqn.py:67-71)
        env.step(choice, pkg.STEP_CHOICE, auto_reset=True)
        return total

    iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows_total = 0
    for _ in range(a.iters):
        rows_total += iteration()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = env.stats()
    print(f"tables={T} iters={a.iters}: {dt / a.iters * 1e3:.1f} ms/iteration, {T * a.iters / dt / 1e6:.2f} M env steps/s, "
          f"{rows_total / a.iters / T:.2f} legal rows per table, {rows_total / dt / 1e6:.2f} M Q evaluations/s, "
          f"episodes={st['episodes']} status={env.status()}")


if __name__ == "__main__":
    main()
