"""Profiling driver: the slab API loop (step_slab(choice)) for T tables, N iterations."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
pkg = importlib.import_module("doudizhu-rl_amd")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
env = pkg.BatchedEnv(T, seed=0, want_ids=False)
env.reset()
env.legal_slab()
choice = torch.zeros(T, dtype=torch.int32, device="cuda")
RANDOM = len(sys.argv) > 3 and sys.argv[3] == "random"  # uniformly random legal moves instead of list entry 0
if RANDOM:
    env.rollout_random(40)
    env.legal_slab()
for _ in range(N):
    if RANDOM:
        choice = (torch.rand(T, device="cuda") * env.counts).to(torch.int32)
    env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)
torch.cuda.synchronize()
print(env.stats(), env.status())
