set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/gpu_tests_8.log 2>&1; echo "tests rc=$?" ; tail -12 gpurun_out/gpu_tests_8.log
python __graft_entry__.py smoke > gpurun_out/smoke_8.log 2>&1; tail -1 gpurun_out/smoke_8.log
bash tools/profile.sh bench > gpurun_out/prof_bench.log 2>&1; echo "bench pass rc=$?"; tail -3 gpurun_out/prof_bench.log
python - <<'PY'
import json
j=json.loads(open('gpurun_out/prof/bench/bench.json').read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], j["roofline"]["issue"]["frac"])
for k,v in j["configs"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(str,dict))})
c=j["cpu_baseline"]; print(c["value"], c["cores"], c["threads_sweep"], c["cgroup_cpu_quota"])
PY
