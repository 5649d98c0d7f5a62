"""Soak of the needed-rows Q forward (csrc/ddz_qnet.h) inside its loop: T tables x N iterations of PolicyLoop (greedy), and
in EVERY iteration the q values of a rotating slice of tables against the LITERAL nn.Conv2d network (net.py:81-102) on the CPU
(fp32, tolerance 1e-5), the need layout against the torch statement, invariants of the segment table, status 0.
  python tools/soak_qnet.py [T=8192] [N=400] [slice=192]"""
import copy
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("doudizhu-rl_amd")
glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
S = int(sys.argv[3]) if len(sys.argv) > 3 else 192
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = glue.QNet(6).to(dev).eval()
net_cpu = copy.deepcopy(net).cpu()
env = pkg.BatchedEnv(T, seed=123, device=dev)
env.reset()
loop = glue.PolicyLoop(env, net, face_variant=3, epsilon=0.05)
fq = loop.fq
worst, rows_checked, t0 = 0.0, 0, time.time()
for it in range(N):
    q = loop.q_values()
    w = fq._ws[("needed", dev, T)]
    seg = w["seg"].cpu().tolist()
    tile = glue.fc_tile()
    assert seg[33] == 0 and all(seg[r] % tile == 0 and seg[r] <= seg[r + 1] for r in range(15)) and seg[15] <= w["cap"], (it, seg)
    lo = (it * S) % (T - S)
    counts = env.counts[lo:lo + S].long().cpu()
    rows = env.slab_rows()[lo:lo + S].cpu()
    face = loop.face[lo:lo + S].cpu()
    qs = q[lo:lo + S].cpu()
    idx = [(t, j) for t in range(S) for j in range(int(counts[t]))]
    tt = torch.tensor([a for a, _ in idx]); jj = torch.tensor([b for _, b in idx])
    acts = (rows[tt, jj, :15].float()[:, :, None] > torch.arange(4)[None, None, :]).float()
    with torch.no_grad():
        want = net_cpu(face[tt], acts)[:, 0]
    err = float((qs[tt, jj] - want).abs().max())
    worst = max(worst, err)
    rows_checked += len(idx)
    assert err < 1e-5, (it, err)
    # every needed (rank, count) of the slice has a row, nothing else
    ri = w["row_index"][lo:lo + S].cpu()
    cnt = rows[tt, jj, :15].long().clamp(0, 4)
    cnt[:, 13:] = cnt[:, 13:].clamp(max=1)
    need = torch.zeros((S, 64), dtype=torch.bool)
    for r in range(15):
        m = cnt[:, r] > 0
        col = (4 * r + cnt[:, r] - 1) if r < 13 else torch.full_like(cnt[:, r], 52 + r - 13)
        need[tt[m], col[m]] = True
    assert bool(((ri >= 0) == need).all()), it
    loop.step()
    if it % 50 == 49:
        print(f"  iteration {it + 1}: worst |q - literal network| so far {worst:.2e}, {rows_checked} moves checked, {time.time() - t0:.0f} s", flush=True)
assert env.status() == 0
st = env.stats()
print(f"soak_qnet: {T} tables x {N} iterations (epsilon 0.05), {st['episodes']} episodes; {rows_checked} moves of rotating {S}-table slices "
      f"against the literal network: worst |dq| {worst:.2e} (tolerance 1e-5); need layout == need sets every iteration; status 0")
