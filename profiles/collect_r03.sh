#!/bin/bash
# Round-3 evidence: run on the GPU box (gpurun -- 'bash profiles/collect_r03.sh'); summaries land in gpurun_out/r3ev/ and
# are copied into profiles/ (r03_*) afterwards.  Counters in passes of their own (never with a trace domain).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ev; mkdir -p $O
for w in c2 c3 c4 c4sv readme; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks_$w -o ks -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/ks_$w.log 2>&1
  python3 profiles/summarize_rocprof.py stats $O/ks_$w/ks_results.db > $O/kernel_stats_$w.txt 2>&1
  rm -rf $O/ks_$w
done
for w in c2 c3 c4 c4sv; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$w -o f -- python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_fetch_$w.log 2>&1
  python3 profiles/summarize_rocprof.py pmc $O/pmc_fetch_$w/f_results.db > $O/pmc_fetch_$w.txt 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$w -o f -- python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_write_$w.log 2>&1
  python3 profiles/summarize_rocprof.py pmc $O/pmc_write_$w/f_results.db > $O/pmc_write_$w.txt 2>&1
  rm -rf $O/pmc_fetch_$w $O/pmc_write_$w
done
grep -h "k_rewrite" $O/pmc_*.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 16 > $O/cli_profile.txt 2>&1
python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 6 -- args -sn 0.005 -in 0.001 -inmin 1 -inmax 50 -de 0.001 -demin 1 -demax 50 -du 0.0005 -dumin 50 -dumax 500 -iv 0.0005 -ivmin 50 -ivmax 500 >> $O/cli_profile.txt 2>&1
# the reference's own benchmark flags (README "Performance": every type at 0.01, translocations included), and the IT pass
python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 6 -- args -sn 0.01 -in 0.01 -de 0.01 -du 0.01 -iv 0.01 -tl 0.01 >> $O/cli_profile.txt 2>&1
python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 6 -- it 0.0001 >> $O/cli_profile.txt 2>&1
# the README's performance table (README.md:430-446: one random sequence of 1 / 10 / 100 / 1000 Mbp, the flags above), one process each
(for mb in 1 10 100 1000; do TMPDIR=/dev/shm python3 mutation-simulator_amd/tools/cli_profile.py --mb $mb --contigs 1 --top 1 -- args -sn 0.01 -in 0.01 -de 0.01 -du 0.01 -iv 0.01 -tl 0.01 2>&1 | grep "CLI wall"; done) > $O/readme_table.txt 2>&1      # (lines: 1, 10, 100, 1000 Mbp; profiles/r03_readme_table.txt adds the labels)
(MSIM_BATCH_PROF=1 python3 mutation-simulator_amd/tools/scaffold_bench.py 20000 10000; python3 mutation-simulator_amd/tools/scaffold_bench.py 200000 1000) > $O/scaffold_bench.txt 2>&1
timeout 100 python3 mutation-simulator_amd/tools/apply_microbench.py 10 > $O/apply_microbench.txt 2>&1
head -30 $O/kernel_stats_c4sv.txt
