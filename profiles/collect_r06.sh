#!/bin/bash
# Round-6 evidence: run on the GPU box (gpurun -- 'bash profiles/collect_r06.sh'); summaries land in gpurun_out/r6ev/ and are copied
# into profiles/ (r06_*) afterwards.  Counters in passes of their own (never with a trace domain); the program itself follows `--`
# (python3 <script>), never a shell or a launcher; every profiled run under a hard timeout of its own.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ev; mkdir -p $O
T="timeout -s KILL 240"
# ---- 1. kernel traces: compatible mode (bench workloads) and the counter-based engine (tools/fast_steps.py)
for w in c2 c3 c4 c4sv; do
  $T rocprofv3 --kernel-trace --stats -d $O/ks_$w -o ks -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/ks_$w.log 2>&1
  python3 profiles/summarize_rocprof.py stats $O/ks_$w/ks_results.db > $O/kernel_stats_$w.txt 2>&1
  [ $w = c2 ] && python3 profiles/summarize_rocprof.py timeline $O/ks_$w/ks_results.db -4 330 1 > $O/timeline_c2.txt 2>&1
  rm -rf $O/ks_$w
  $T rocprofv3 --kernel-trace --stats -d $O/kf_$w -o ks -- python3 mutation-simulator_amd/tools/fast_steps.py $w 3 > $O/kf_$w.log 2>&1
  (grep "plan+apply\|plan only\|host enqueue" $O/kf_$w.log; python3 profiles/summarize_rocprof.py stats $O/kf_$w/ks_results.db) > $O/kernel_stats_fast_$w.txt 2>&1
  [ $w = c3 ] && python3 profiles/summarize_rocprof.py timeline $O/kf_$w/ks_results.db -6 120 2 k_fsplit_top > $O/timeline_fast_c3.txt 2>&1
  rm -rf $O/kf_$w
done
# ---- 2. HBM traffic of the rewrite kernels (FETCH_SIZE / WRITE_SIZE, separate passes)
for w in c2 c3 c4 c4sv; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    n=$(echo $ctr | tr A-Z a-z | sed 's/_size//')
    MSIM_BENCH_NO_PMC=1 $T rocprofv3 --pmc $ctr -d $O/pmc_${n}_$w -o f -- python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_${n}_$w.log 2>&1
    python3 profiles/summarize_rocprof.py pmc $O/pmc_${n}_$w/f_results.db > $O/pmc_${n}_$w.txt 2>&1
    rm -rf $O/pmc_${n}_$w
  done
done
grep -h "k_rewrite" $O/pmc_*.txt
# ---- 3. SQ counters of k_rewrite<140> on the SV mix, one 240 Mb contig, two passes
$T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/sq1 -o s -- python3 mutation-simulator_amd/tools/apply_microbench.py 4 > $O/sq1.log 2>&1
$T rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM -d $O/sq2 -o s -- python3 mutation-simulator_amd/tools/apply_microbench.py 4 > $O/sq2.log 2>&1
(python3 profiles/summarize_rocprof.py pmc $O/sq1/s_results.db; python3 profiles/summarize_rocprof.py pmc $O/sq2/s_results.db) 2>&1 | grep "k_rewrite" > $O/pmc_sq_k_rewrite.txt
rm -rf $O/sq1 $O/sq2
# ---- 4. the text kernels through the CLI: kernel trace + the sizes that turn durations into GB/s
for tag in c2 c3 readme; do
  case $tag in
    c2) flags="args -sn 0.01 -titv 2.0";;
    c3) flags="args -sn 0.005 -in 0.001 -inmin 1 -inmax 50 -de 0.001 -demin 1 -demax 50 -du 0.0005 -dumin 50 -dumax 500 -iv 0.0005 -ivmin 50 -ivmax 500";;
    readme) flags="args -sn 0.01 -in 0.01 -de 0.01 -du 0.01 -iv 0.01 -tl 0.01";;
  esac
  $T rocprofv3 --kernel-trace --stats -d $O/kc_$tag -o ks -- python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 4 -- $flags > $O/kc_$tag.log 2>&1
  (grep "CLI wall\|input:" $O/kc_$tag.log; python3 profiles/summarize_rocprof.py text $O/kc_$tag/ks_results.db "$(grep 'CLI wall' $O/kc_$tag.log)") > $O/kernel_stats_cli_$tag.txt 2>&1
  rm -rf $O/kc_$tag
done
# ---- 5. the driver's command, CLI profiles (no profiler attached), micro-benchmarks
$T python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
$T python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 12 > $O/cli_profile.txt 2>&1
$T python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 6 -- args -sn 0.005 -in 0.001 -inmin 1 -inmax 50 -de 0.001 -demin 1 -demax 50 -du 0.0005 -dumin 50 -dumax 500 -iv 0.0005 -ivmin 50 -ivmax 500 >> $O/cli_profile.txt 2>&1
$T python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 6 -- args -sn 0.01 -in 0.01 -de 0.01 -du 0.01 -iv 0.01 -tl 0.01 >> $O/cli_profile.txt 2>&1
$T python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 6 -- --rng fast args -sn 0.005 -in 0.001 -inmin 1 -inmax 50 -de 0.001 -demin 1 -demax 50 -du 0.0005 -dumin 50 -dumax 500 -iv 0.0005 -ivmin 50 -ivmax 500 >> $O/cli_profile.txt 2>&1
$T python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 20000 --top 6 >> $O/cli_profile.txt 2>&1
$T python3 mutation-simulator_amd/tools/cli_profile.py --mb 1200 --contigs 6 --top 6 -- it 0.000001 >> $O/cli_profile.txt 2>&1
timeout -s KILL 100 python3 mutation-simulator_amd/tools/apply_microbench.py 10 > $O/apply_microbench.txt 2>&1
( for i in 1 2 3; do echo "# round 6 defaults"; $T python3 mutation-simulator_amd/tools/compat_steps.py c2 20; echo "# MSIM_NO_AHEAD=1 (round 5's schedule for a rank that owns everything)"; MSIM_NO_AHEAD=1 $T python3 mutation-simulator_amd/tools/compat_steps.py c2 20; done ) > $O/sharded_rank_steps.txt 2>&1
MSIM_BENCH_DEVICE=0 MSIM_BENCH_RCCL_TIMEOUT=60 $T python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_n2_one_gpu.json 2> $O/bench_n2_one_gpu.err; echo "rc=$?" >> $O/bench_n2_one_gpu.err
python3 profiles/make_traffic.py $O $O/bench_default.json > $O/traffic.json 2> $O/traffic.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r6ev/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["stages_ms_per_step"], d["cpu_baseline"].get("matches_gpu"))
s=d["secondary"]
for k in ("c3","c4","c4sv"): print(k, s[k]["value"], s[k]["ms_per_step"], s[k]["roofline"]["frac"], s[k]["stages_ms_per_step"])
f=s["fast_rng"]
for k in ("c2","c3","c4","c4sv"): print("fast",k,f[k]["ms_per_step"], f[k]["step_roofline"]["frac"], f[k]["rewrite_kernel_frac_of_hbm_peak"])
print({k:(s[k]["value"], s[k]["wall_s"]) for k in s if k.startswith("e2e")})
PY
grep "CLI wall" $O/cli_profile.txt
cat $O/apply_microbench.txt
tail -3 $O/kernel_stats_cli_c3.txt
