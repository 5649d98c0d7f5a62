"""Timing A/B of the one-table-per-wave form of k_slab (T <= 4096): step_slab(RANDOM) in the steady state, microseconds per
launch by HIP events over N back-to-back launches, for
  team   the lists of plane-rich leads written by the whole block (the default),
  single every list by its table's wave (ddz_debug_set_geometry slab_coop = 2),
  tpw2   two tables per wave (the work-list form: half the blocks).
  python tools/slab_coop_probe.py [T=4096] [N=2000] [ids=1]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
pkg = importlib.import_module("doudizhu-rl_amd")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
IDS = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
for name, kw in (("team5", {}), ("single", {"_debug_slab_coop": 2}), ("team3", {"_debug_slab_coop": 3}), ("team8", {"_debug_slab_coop": 8}), ("team16", {"_debug_slab_coop": 16}), ("tpw2", {"_debug_tables_per_wave": 2})):
    env = pkg.BatchedEnv(T, seed=0, want_ids=IDS, **kw)
    env.reset()
    env.rollout_random(200)
    env.legal_slab()
    for _ in range(50):
        env.step_slab(None, pkg.STEP_RANDOM, auto_reset=True)
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            env.step_slab(None, pkg.STEP_RANDOM, auto_reset=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / N
        best = us if best is None or us < best else best
    # one launch at a time (launch + its completion): what a caller that reads the lists every ply sees is not this, but
    # the spread between back-to-back and isolated launches shows the cold-start share
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iso = []
    for _ in range(200):
        torch.cuda.synchronize()
        e0.record()
        env.step_slab(None, pkg.STEP_RANDOM, auto_reset=True)
        e1.record()
        torch.cuda.synchronize()
        iso.append(e0.elapsed_time(e1) * 1e3)
    iso.sort()
    print(f"T={T} ids={int(IDS)} {name:7s} {best:7.2f} us per launch back to back   isolated: median {iso[100]:6.2f}  p90 {iso[180]:6.2f}   status {env.status()}")
    del env
