#!/bin/bash
# Round-4 evidence: run on the GPU box (gpurun -- 'bash profiles/collect_r04.sh'); summaries land in gpurun_out/r4ev/ and are
# copied into profiles/ (r04_*) afterwards.  Counters in passes of their own (never with a trace domain); the program itself
# follows `--` (python3 <script>), never a shell or a launcher.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ev; mkdir -p $O
# ---- 1. kernel traces: compatible mode (bench workloads) and the counter-based engine (tools/fast_steps.py)
for w in c2 c3 c4 c4sv; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks_$w -o ks -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/ks_$w.log 2>&1
  python3 profiles/summarize_rocprof.py stats $O/ks_$w/ks_results.db > $O/kernel_stats_$w.txt 2>&1
  [ $w = c2 ] && python3 profiles/summarize_rocprof.py timeline $O/ks_$w/ks_results.db -4 330 2 > $O/timeline_c2.txt 2>&1
  rm -rf $O/ks_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kf_$w -o ks -- python3 mutation-simulator_amd/tools/fast_steps.py $w 3 > $O/kf_$w.log 2>&1
  (grep "plan+apply\|plan only\|host enqueue" $O/kf_$w.log; python3 profiles/summarize_rocprof.py stats $O/kf_$w/ks_results.db) > $O/kernel_stats_fast_$w.txt 2>&1
  [ $w = c3 ] && python3 profiles/summarize_rocprof.py timeline $O/kf_$w/ks_results.db -6 120 2 k_fsplit_top > $O/timeline_fast_c3.txt 2>&1
  rm -rf $O/kf_$w
done
# ---- 2. HBM traffic of the rewrite kernels (FETCH_SIZE / WRITE_SIZE, separate passes)
for w in c2 c3 c4 c4sv; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    n=$(echo $ctr | tr A-Z a-z | sed 's/_size//')
    MSIM_BENCH_NO_PMC=1 timeout 300 rocprofv3 --pmc $ctr -d $O/pmc_${n}_$w -o f -- python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_${n}_$w.log 2>&1
    python3 profiles/summarize_rocprof.py pmc $O/pmc_${n}_$w/f_results.db > $O/pmc_${n}_$w.txt 2>&1
    rm -rf $O/pmc_${n}_$w
  done
done
grep -h "k_rewrite" $O/pmc_*.txt
# ---- 3. SQ counters of k_rewrite<140> on the SV mix (what bounds it: VALU issue, LDS, waiting?), one 240 Mb contig, two passes
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/sq1 -o s -- python3 mutation-simulator_amd/tools/apply_microbench.py 4 > $O/sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM -d $O/sq2 -o s -- python3 mutation-simulator_amd/tools/apply_microbench.py 4 > $O/sq2.log 2>&1
(python3 profiles/summarize_rocprof.py pmc $O/sq1/s_results.db; python3 profiles/summarize_rocprof.py pmc $O/sq2/s_results.db) 2>&1 | grep "k_rewrite" > $O/pmc_sq_k_rewrite.txt
rm -rf $O/sq1 $O/sq2
# ---- 4. the text kernels through the CLI: kernel trace + the sizes that turn durations into GB/s
for tag in c2 c3 readme; do
  case $tag in
    c2) flags="args -sn 0.01 -titv 2.0";;
    c3) flags="args -sn 0.005 -in 0.001 -inmin 1 -inmax 50 -de 0.001 -demin 1 -demax 50 -du 0.0005 -dumin 50 -dumax 500 -iv 0.0005 -ivmin 50 -ivmax 500";;
    readme) flags="args -sn 0.01 -in 0.01 -de 0.01 -du 0.01 -iv 0.01 -tl 0.01";;
  esac
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kc_$tag -o ks -- python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 4 -- $flags > $O/kc_$tag.log 2>&1
  (grep "CLI wall\|input:" $O/kc_$tag.log; python3 profiles/summarize_rocprof.py text $O/kc_$tag/ks_results.db "$(grep 'CLI wall' $O/kc_$tag.log)") > $O/kernel_stats_cli_$tag.txt 2>&1
  rm -rf $O/kc_$tag
done
# ---- 5. the driver's command, CLI profiles, micro-benchmarks
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 12 > $O/cli_profile.txt 2>&1
python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 6 -- args -sn 0.005 -in 0.001 -inmin 1 -inmax 50 -de 0.001 -demin 1 -demax 50 -du 0.0005 -dumin 50 -dumax 500 -iv 0.0005 -ivmin 50 -ivmax 500 >> $O/cli_profile.txt 2>&1
python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 6 -- --rng fast args -sn 0.005 -in 0.001 -inmin 1 -inmax 50 -de 0.001 -demin 1 -demax 50 -du 0.0005 -dumin 50 -dumax 500 -iv 0.0005 -ivmin 50 -ivmax 500 >> $O/cli_profile.txt 2>&1
python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 20000 --top 6 >> $O/cli_profile.txt 2>&1
python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 6 -- it 0.000001 >> $O/cli_profile.txt 2>&1
MSIM_IO_PROF=1 python3 mutation-simulator_amd/tools/cli_profile.py --dir /dev/shm --mb 1200 --contigs 6 --top 2 2>&1 | grep "msim io\|CLI wall" > $O/file_channels.txt
timeout 100 python3 mutation-simulator_amd/tools/apply_microbench.py 10 > $O/apply_microbench.txt 2>&1
# ---- 6. why the output channels write() instead of mapping the files (csrc/file_io.hip)
g++ -O2 -pthread -o /tmp/iobench mutation-simulator_amd/tools/iobench.cpp 2>/dev/null
(echo "== /dev/shm (tmpfs)"; /tmp/iobench /dev/shm | tail -17; echo "== /tmp"; /tmp/iobench /tmp | tail -17) > $O/iobench.txt 2>&1
head -12 $O/kernel_stats_fast_c3.txt
