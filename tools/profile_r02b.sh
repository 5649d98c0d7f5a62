# Round-2 evidence pass B (GPU box): config 4 (rule-based opponent) example + kernel stats of k_auto.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02b
mkdir -p $O
timeout -k 10 500 python examples/config4_rule_opponent.py --tables 65536 --iters 100 > $O/config4_random.txt 2>&1
cat $O/config4_random.txt
timeout -k 10 300 python examples/config4_rule_opponent.py --tables 4096 --iters 200 > $O/config4_random_4096.txt 2>&1
cat $O/config4_random_4096.txt
timeout -k 10 500 python examples/config4_rule_opponent.py --tables 65536 --iters 20 --lord net > $O/config4_net.txt 2>&1
cat $O/config4_net.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/config4_stats -o p -- python3 examples/config4_rule_opponent.py --tables 65536 --iters 40 > $O/config4_stats.log 2>&1
head -8 $O/config4_stats/p_kernel_stats.csv
