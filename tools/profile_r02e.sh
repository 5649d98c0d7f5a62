# Round-2 evidence pass E (GPU box): the lane-parallel k_slab -- instruction counters + kernel stats (same commands as
# profile_r02c.sh's slab legs, so that tools/collect_r02.py picks them up).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02c
mkdir -p $O
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY"
P2="SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
CMD="python3 tools/slab_loop.py 65536 200"
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $O/slab_p1 -o p -- $CMD > $O/slab_p1.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $O/slab_p2 -o p -- $CMD > $O/slab_p2.log 2>&1
echo slab pmc done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab_stats -o p -- python3 tools/slab_loop.py 65536 200 > $O/slab_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab_stats_4096 -o p -- python3 tools/slab_loop.py 4096 400 > $O/slab_stats_4096.log 2>&1
head -4 $O/slab_stats/p_kernel_stats.csv; head -4 $O/slab_stats_4096/p_kernel_stats.csv
