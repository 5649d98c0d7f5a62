# Round-2 evidence pass A (GPU box): VALU issue probe + rocprofv3 summaries of the secondary kernels.
# Run: gpurun --timeout 1100 -- 'bash tools/profile_r02a.sh'
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02a
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip
/tmp/valu_issue_probe > $O/valu_issue_probe.txt 2>&1
tail -n 5 $O/valu_issue_probe.txt
# real shader clock + issued / active VALU cycles of the same probe launches
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/probe_pmc -o p -- /tmp/valu_issue_probe > $O/probe_pmc.log 2>&1
echo probe pmc done
# secondary kernels: kernel stats, then HBM counters in separate passes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/observe_stats -o p -- python3 tools/observe_probe.py 65536,524288 > $O/observe_stats.log 2>&1
echo observe stats done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/observe_pmc_$c -o p -- python3 tools/observe_probe.py 524288 > $O/observe_pmc_$c.log 2>&1
done
echo observe pmc done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/moves_stats -o p -- python3 tools/get_moves_probe.py > $O/moves_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/moves_pmc_$c -o p -- python3 tools/get_moves_probe.py > $O/moves_pmc_$c.log 2>&1
done
echo moves done
find $O -name "*.csv" | head -40
