# Round-2 evidence pass F (GPU box): instruction counters of k_auto2 (branch and bound) on the config-4 example.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02f
mkdir -p $O
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY"
P2="SQ_INST_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
CMD="python3 examples/config4_rule_opponent.py --tables 65536 --iters 30"
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $O/auto_p1 -o p -- $CMD > $O/auto_p1.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $O/auto_p2 -o p -- $CMD > $O/auto_p2.log 2>&1
echo auto pmc done
