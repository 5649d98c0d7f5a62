set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slab_api or large_batch" > gpurun_out/gpu_tests_14.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/gpu_tests_14.log
python - <<'PY'
import importlib, sys, time, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("doudizhu-rl_amd")
for T in (65536, 16384, 32768, 131072):
  for wl in (0, 1):
    env = pkg.BatchedEnv(T, seed=0, _debug_slab_work_list=wl)
    env.reset(); env.rollout_random(200); env.legal_slab()
    choice = torch.zeros(T, dtype=torch.int32, device="cuda")
    out = []
    for name, fn in (("random", lambda: env.step_slab(None, pkg.STEP_RANDOM)), ("choice0", lambda: env.step_slab(choice, pkg.STEP_CHOICE))):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 300
        for _ in range(n): fn()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / n * 1e6)
    print(f"T={T} work_list={wl}: random {out[0]:7.1f} us  choice(entry 0) {out[1]:7.1f} us  status {env.status()}", flush=True)
    del env
PY
