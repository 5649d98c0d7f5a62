# Round-3 evidence pass A (GPU box): k_slab in the modes of bench.py's legs -- kernel stats + two PMC passes.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a
mkdir -p $O
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY"
P2="SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"
for T in 65536 4096; do
  for m in random fused; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${m}_${T}_stats -o p -- python3 tools/slab_modes_probe.py $T 200 $m > $O/${m}_${T}_stats.log 2>&1
    rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $O/${m}_${T}_p1 -o p -- python3 tools/slab_modes_probe.py $T 100 $m > $O/${m}_${T}_p1.log 2>&1
    rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $O/${m}_${T}_p2 -o p -- python3 tools/slab_modes_probe.py $T 100 $m > $O/${m}_${T}_p2.log 2>&1
    echo $m $T done
    head -3 $O/${m}_${T}_stats/p_kernel_stats.csv
  done
done
