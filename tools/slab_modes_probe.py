"""Profiling driver: the slab stepping launch in the modes bench.py's legs use (ids on, as the default BatchedEnv):
  python tools/slab_modes_probe.py T N random|fused|choice
random = step_slab(RANDOM) -> k_slab<0,true>; fused = policy_step_slab(random q, face = EnvCooperationSimplify) ->
k_slab<4,true>; choice = step_slab(CHOICE) with a uniformly random legal index -> k_slab<1,true>.  Put it after `--` of rocprofv3."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
pkg = importlib.import_module("doudizhu-rl_amd")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = sys.argv[3] if len(sys.argv) > 3 else "random"
env = pkg.BatchedEnv(T, seed=0)
env.reset()
env.rollout_random(200)
env.legal_slab()
face = torch.empty((T, 6, 15, 4), dtype=torch.float32, device="cuda")
q = torch.rand((T, env.slab_stride), dtype=torch.float32, device="cuda")
s0 = env.stats()
for _ in range(N):
    if mode == "random":
        env.step_slab(None, pkg.STEP_RANDOM, auto_reset=True)
    elif mode == "fused":
        env.policy_step_slab(q, 0.0, face_variant=3, face_out=face)
    else:
        choice = (torch.rand(T, device="cuda") * env.counts).to(torch.int32)
        env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)
torch.cuda.synchronize()
s1 = env.stats()
print(mode, T, N, {k: s1[k] - s0[k] for k in s1}, env.status())
