set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "q_slab or q_network" > gpurun_out/gpu_tests_18.log 2>&1; echo "tests rc=$?" ; tail -2 gpurun_out/gpu_tests_18.log
bash tools/profile.sh dqn rollout > gpurun_out/prof_18.log 2>&1; echo "profile rc=$?"; grep -E "k_q_feat|Cijk_Ailk_Bljk_S_B_Bias_HA_S_SAV_UserArgs_MT256x256|k_q_slab" gpurun_out/prof/dqn/stats/p_kernel_stats.csv | cut -c1-60,500-; tail -1 gpurun_out/prof/dqn/config3.txt
