set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "policy or graph or tiny or q_network" > gpurun_out/gpu_tests_15.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/gpu_tests_15.log
python - <<'PY'
import importlib, sys, time, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("doudizhu-rl_amd")
for T in (65536, 16384):
  for wl in (0, 1):
    env = pkg.BatchedEnv(T, seed=0, _debug_slab_work_list=wl)
    env.reset(); env.rollout_random(200); env.legal_slab()
    q = torch.rand((T, env.slab_stride), device="cuda"); face = torch.empty((T, 6, 15, 4), device="cuda")
    out = []
    for name, fn in (("fused", lambda: env.policy_step_slab(q, 0.0, face_variant=3, face_out=face)), ("fused_noface", lambda: env.policy_step_slab(q, 0.0))):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): fn()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 300 * 1e6)
    print(f"T={T} work_list={wl}: fused {out[0]:7.1f} us  fused without face {out[1]:7.1f} us  status {env.status()}", flush=True)
    del env, q, face
PY
