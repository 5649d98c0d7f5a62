mkdir -p gpurun_out
(python tools/soak.py 16384 3000 1000 2>&1 | grep -v amdgpu.ids; python tools/soak.py 65536 400 120 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/soak_r03.txt
