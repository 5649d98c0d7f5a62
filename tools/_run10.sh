mkdir -p gpurun_out
bash tools/profile.sh slab auto dqn probe stamps > gpurun_out/prof_all.log 2>&1; echo "profile rc=$?"; tail -40 gpurun_out/prof_all.log
