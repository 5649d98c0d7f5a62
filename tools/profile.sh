#!/bin/bash
# Evidence passes on the GPU box (rocprofv3; the program always directly after `--`, counters in their own runs).
#   gpurun --timeout 1100 -- 'bash tools/profile.sh <pass> [<pass> ...]'        raw output: gpurun_out/prof/<pass>/
# passes:
#   bench      bench.py + its --kernel-trace --stats summary (headline kernel k_rollout and every leg)
#   rollout    k_rollout: HBM counters (FETCH_SIZE / WRITE_SIZE, separate passes) + instruction counters at 4096 tables
#   slab       k_slab in the modes of bench.py's legs (step_slab(RANDOM) = k_slab<0,true>, fused policy step = k_slab<4,true>)
#              at 65,536 and 4096 tables: kernel stats + two PMC passes (instructions / waits, LDS / issue)
#   auto       the rule-agent loop (examples/config4_rule_opponent.py): kernel stats + two PMC passes of k_auto2
#   dqn        configs[2] with the Q-network in the loop (examples/config3_dqn_inference.py): kernel stats, MFMA / wait counters,
#              HBM counters (separate passes), per-stage HIP-event times
#   secondary  k_observe<0..3>, k_mask, k_moves*: kernel stats + HBM counters
#   probe      tools/valu_issue_probe.hip (VALU issue rates) + tools/dpp_probe.hip (DPP scan / reduction vs ds_bpermute)
#   stamps     -DDDZ_STAMP builds: where k_slab's and k_auto2's waves spend their cycles (tools/stamp_slab.py, stamp_auto.py)
# tools/collect_r04.py turns the raw output into the summaries tracked under profiles/ (tools/README.md maps each file).
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY"
P2="SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"
stats() { d=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- "$@" > "$d.log" 2>&1; }
pmc() { d=$1; c=$2; shift 2; rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d" -o p -- "$@" > "$d.log" 2>&1; }
for pass in "$@"; do
  O=gpurun_out/prof/$pass
  mkdir -p "$O"
  case $pass in
    bench)
      python3 bench.py > $O/bench.json 2> $O/bench.err
      stats $O/bench_stats python3 bench.py --no-cpu-baseline
      head -8 $O/bench_stats/p_kernel_stats.csv ;;
    rollout)
      for c in FETCH_SIZE WRITE_SIZE; do pmc $O/pmc_$c $c python3 tools/run_rollout.py 4096 2000; done
      pmc $O/pmc_mix "$P1" python3 tools/run_rollout.py 4096 20000
      # the headline configuration (65,536 tables): the same three passes
      for c in FETCH_SIZE WRITE_SIZE; do pmc $O/big_pmc_$c $c python3 tools/run_rollout.py 65536 500; done
      pmc $O/big_pmc_mix "$P1" python3 tools/run_rollout.py 65536 2000 ;;
    slab)
      for T in 65536 4096; do
        for m in random fused; do
          stats $O/${m}_${T}_stats python3 tools/slab_modes_probe.py $T 200 $m
          pmc $O/${m}_${T}_p1 "$P1" python3 tools/slab_modes_probe.py $T 100 $m
          pmc $O/${m}_${T}_p2 "$P2" python3 tools/slab_modes_probe.py $T 100 $m
          head -2 $O/${m}_${T}_stats/p_kernel_stats.csv
        done
      done ;;
    auto)
      CMD="python3 examples/config4_rule_opponent.py --tables 65536 --iters 40"
      stats $O/stats $CMD
      pmc $O/p1 "$P1" $CMD
      pmc $O/p2 "$P2" $CMD
      python3 examples/config4_rule_opponent.py --tables 65536 --iters 100 > $O/config4_random_65536.txt 2>&1
      python3 examples/config4_rule_opponent.py --tables 65536 --iters 20 --lord net > $O/config4_net_65536.txt 2>&1
      python3 examples/config4_rule_opponent.py --tables 4096 --iters 200 > $O/config4_random_4096.txt 2>&1
      head -4 $O/stats/p_kernel_stats.csv; cat $O/config4_*.txt ;;
    dqn)
      stats $O/stats python3 examples/config3_dqn_inference.py --iters 10
      pmc $O/p_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" python3 examples/config3_dqn_inference.py --iters 8
      pmc $O/p_mfma_k_fc1 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" python3 examples/config3_dqn_inference.py --iters 8 --gemm mfma
      for c in FETCH_SIZE WRITE_SIZE; do pmc $O/pmc_$c $c python3 examples/config3_dqn_inference.py --iters 6; done
      python3 examples/config3_dqn_inference.py --iters 20 --stages > $O/config3.txt 2>&1
      head -14 $O/stats/p_kernel_stats.csv; tail -3 $O/config3.txt ;;
    secondary)
      stats $O/observe_stats python3 tools/observe_probe.py 65536,524288
      for c in FETCH_SIZE WRITE_SIZE; do pmc $O/observe_pmc_$c $c python3 tools/observe_probe.py 524288; done
      stats $O/moves_stats python3 tools/get_moves_probe.py
      for c in FETCH_SIZE WRITE_SIZE; do pmc $O/moves_pmc_$c $c python3 tools/get_moves_probe.py; done ;;
    probe)
      hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip && /tmp/valu_issue_probe > $O/valu_issue_probe.txt 2>&1
      hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_probe tools/dpp_probe.hip 2> /dev/null && /tmp/dpp_probe > $O/dpp_probe.txt 2>&1
      tail -5 $O/valu_issue_probe.txt; cat $O/dpp_probe.txt ;;
    stamps)
      python3 tools/stamp_slab.py 65536,4096 random > $O/stamp_slab.txt 2>&1
      python3 tools/stamp_auto.py 16384 > $O/stamp_auto.txt 2>&1
      cat $O/stamp_slab.txt $O/stamp_auto.txt ;;
    *) echo "unknown pass $pass"; exit 2 ;;
  esac
  echo "pass $pass done"
done
