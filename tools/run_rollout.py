"""Profiling driver: `python tools/run_rollout.py T ITERS` runs one seeded random-policy
rollout (no oracle, no CPU baseline); put it after `--` of rocprofv3."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("doudizhu-rl_amd")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
env = pkg.BatchedEnv(T, seed=0, want_ids=False)
env.reset()
env.rollout_random(50)
torch.cuda.synchronize()
env.rollout_random(iters)
torch.cuda.synchronize()
print(T, iters, env.stats(), env.status())
