"""How much do k_auto2's teams speed up a HEAVY decision?  T tables of which the first NA are the heaviest cases of fixture G8h
(the others are finished tables), decided with teams off (0), team-first (1: the block's waves help from the first trip, the
owner's frontier shared out through the box) and end-of-queue teams (2).  With NA <= 256 every decision has a wave (mode 0) or a
block (mode 1) of its own: the launch lasts as long as its slowest decision, so t(0) / t(1) is the team's speed-up on exactly
the searches that make the tail of a 65,536-table launch.
  python tools/team_probe.py [T=2048] [NA=T]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def heavy_states(pkg, T, NA):
    """(state bytes uint8 [T,11,16], expected ids int32 [T]): G8h's states, heaviest first, as tables (byte 15 of a handout row =
    its category, which the state entry point reads instead of classifying the row again)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "rule_agent_heavy.npz"))
    order = np.argsort(-g["nodes"])
    pick = order[np.arange(T) % len(order)]
    cat = {bytes(r[:15]): int(r[15]) for r in pkg.action_table().cpu().numpy()}
    st = np.zeros((T, 11, 16), np.uint8)
    st[:, 10, 1] = 1
    for i, k in enumerate(pick[:NA]):
        role = int(g["role"][k])
        st[i, role, :15] = g["hand"][k]
        for r in range(3):
            st[i, r, 15] = g["left"][k][r]
        st[i, 6 + (role + 2) % 3, :15] = g["last"][k]
        st[i, 6 + (role + 2) % 3, 15] = cat[bytes(g["last"][k].astype(np.int8))]
        st[i, 10, 0] = role; st[i, 10, 1] = 0; st[i, 10, 2] = 0xFF; st[i, 10, 6] = 1
    want = g["choice"][pick].astype(np.int32)
    want[NA:] = -1
    return st, want, g["nodes"][pick[:NA]]


if __name__ == "__main__":
    import torch
    pkg = importlib.import_module("doudizhu-rl_amd")
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    NA = int(sys.argv[2]) if len(sys.argv) > 2 else T
    st, want, nodes = heavy_states(pkg, T, NA)
    want = torch.from_numpy(want).cuda()
    for mode in (0, 2, 1):
        env = pkg.BatchedEnv(T, seed=0, want_ids=False, _debug_auto_teams=mode)
        env.state_import(torch.from_numpy(st).view(-1))
        ids = env.auto_choose(0b111)
        torch.cuda.synchronize()
        ok = bool((ids == want).all())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); env.auto_choose(0b111); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"T={T} active {NA} teams mode {mode}: {best * 1e3:8.0f} us per launch, ids == reference: {ok}, status {env.status()}; "
              f"full-enumeration nodes of the heaviest {int(nodes.max())}, mean {nodes.mean():.0f}")
