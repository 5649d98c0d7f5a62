# Round-2 evidence pass D (GPU box): final kernel stats of the slab loop, the rule-agent loop and bench.py.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02d
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o p -- python3 bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab_stats -o p -- python3 tools/slab_loop.py 65536 200 > $O/slab_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab_stats_4096 -o p -- python3 tools/slab_loop.py 4096 400 > $O/slab_stats_4096.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/config4_stats -o p -- python3 examples/config4_rule_opponent.py --tables 65536 --iters 40 > $O/config4_stats.log 2>&1
python examples/config4_rule_opponent.py --tables 65536 --iters 100 > $O/config4_random.txt 2>&1
python examples/config4_rule_opponent.py --tables 65536 --iters 20 --lord net > $O/config4_net.txt 2>&1
python examples/config4_rule_opponent.py --tables 4096 --iters 200 > $O/config4_random_4096.txt 2>&1
cat $O/config4_random.txt $O/config4_net.txt $O/config4_random_4096.txt
head -3 $O/slab_stats/p_kernel_stats.csv; head -3 $O/slab_stats_4096/p_kernel_stats.csv; head -4 $O/config4_stats/p_kernel_stats.csv; head -6 $O/bench_stats/p_kernel_stats.csv
