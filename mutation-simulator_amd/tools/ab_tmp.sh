cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/ah; mkdir -p $O
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- python3 mutation-simulator_amd/tools/compat_steps.py c3 3 3e9 1 > $O/ks.log 2>&1
python3 profiles/summarize_rocprof.py timeline $O/ks/ks_results.db -2 700 2 > $O/timeline_c3.txt 2>&1
rm -rf $O/ks
MSIM_WALK_PROF=1 MSIM_CHAIN_PROF=1 timeout -s KILL 120 python3 mutation-simulator_amd/tools/compat_steps.py c3 5 3e9 1 > $O/c3_prof.txt 2>&1
tail -30 $O/c3_prof.txt
