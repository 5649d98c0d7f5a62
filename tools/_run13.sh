set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_13.log 2>&1; echo "tests rc=$?" ; tail -4 gpurun_out/gpu_tests_13.log
bash tools/profile.sh bench slab stamps > gpurun_out/prof_13.log 2>&1; echo "profile rc=$?"; tail -30 gpurun_out/prof_13.log
python - <<'PY'
import json
j=json.loads(open('gpurun_out/prof/bench/bench.json').read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], j["roofline"]["issue"]["frac"])
for k,v in j["configs"].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(str,dict))})
PY
