set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_17.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/gpu_tests_17.log
bash tools/profile.sh bench slab > gpurun_out/prof_17.log 2>&1; echo "profile rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/prof/bench/bench.json').read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], j["roofline"]["issue"]["frac"], j["roofline"]["traffic"])
for k,v in j["configs"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(str,dict))})
print({k: round(v["us_per_call"],1) for k,v in j["configs"]["stress_plane_rich_leads"].items() if isinstance(v, dict)})
c=j["cpu_baseline"]; print(c["value"], c["cores"], c["single_core_value"])
PY
