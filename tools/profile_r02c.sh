# Round-2 evidence pass C (GPU box): instruction counters of k_slab / k_rollout / k_auto + calibration on the VALU probe.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02c
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY"
P2="SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for w in slab rollout; do
  if [ $w = slab ]; then CMD="python3 tools/slab_loop.py 65536 200"; else CMD="python3 tools/run_rollout.py 4096 20000"; fi
  rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $O/${w}_p1 -o p -- $CMD > $O/${w}_p1.log 2>&1
  rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $O/${w}_p2 -o p -- $CMD > $O/${w}_p2.log 2>&1
  echo $w done
done
rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $O/probe_p2 -o p -- /tmp/valu_issue_probe > $O/probe_p2.log 2>&1
echo probe done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab_stats -o p -- python3 tools/slab_loop.py 65536 200 > $O/slab_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab_stats_4096 -o p -- python3 tools/slab_loop.py 4096 400 > $O/slab_stats_4096.log 2>&1
head -4 $O/slab_stats/p_kernel_stats.csv; head -4 $O/slab_stats_4096/p_kernel_stats.csv
