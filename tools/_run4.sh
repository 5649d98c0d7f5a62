set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "slab or graph or policy or q_slab or rule_opponent or smoke" --durations=8 > gpurun_out/gpu_tests_4.log 2>&1; echo "tests rc=$?" ; tail -16 gpurun_out/gpu_tests_4.log
python bench.py --no-cpu-baseline > gpurun_out/bench_4.json 2> gpurun_out/bench_4.err; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_4.json').read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"])
for k,v in j["configs"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(str,dict))})
PY
python tools/stamp_slab.py 65536 random > gpurun_out/stamp_slab_r03b.txt 2>&1; cat gpurun_out/stamp_slab_r03b.txt
python - <<'PY'
import importlib, sys, time, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("doudizhu-rl_amd")
T = 65536
for chunk in (0, 4, 8, 16):
    env = pkg.BatchedEnv(T, seed=0, _debug_slab_chunk=chunk)
    env.reset(); env.rollout_random(200); env.legal_slab()
    q = torch.rand((T, env.slab_stride), device="cuda"); face = torch.empty((T, 6, 15, 4), device="cuda")
    for name, fn in (("random", lambda: env.step_slab(None, pkg.STEP_RANDOM)), ("fused", lambda: env.policy_step_slab(q, 0.0, face_variant=3, face_out=face))):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
        print(f"chunk={chunk:2d} {name:6s} {dt * 1e6:6.1f} us", flush=True)
    del env
PY
