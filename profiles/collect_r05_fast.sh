#!/bin/bash
# Round-5 evidence, the part that changed last (fast mode's leaf size): kernel traces of tools/fast_steps.py and the driver's command.
# Same rules as collect_r05.sh; summaries land in gpurun_out/r5ev/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ev; mkdir -p $O
T="timeout -s KILL 240"
for w in c2 c3 c4 c4sv; do
  $T rocprofv3 --kernel-trace --stats -d $O/kf_$w -o ks -- python3 mutation-simulator_amd/tools/fast_steps.py $w 3 > $O/kf_$w.log 2>&1
  (grep "plan+apply\|plan only\|host enqueue" $O/kf_$w.log; python3 profiles/summarize_rocprof.py stats $O/kf_$w/ks_results.db) > $O/kernel_stats_fast_$w.txt 2>&1
  [ $w = c3 ] && python3 profiles/summarize_rocprof.py timeline $O/kf_$w/ks_results.db -6 120 2 k_fsplit_top > $O/timeline_fast_c3.txt 2>&1
  rm -rf $O/kf_$w
done
$T python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r5ev/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"].get("matches_gpu"))
s=d["secondary"]
for k in ("c3","c4","c4sv"): print(k, s[k]["value"], s[k]["ms_per_step"], s[k]["roofline"]["frac"])
f=s["fast_rng"]
for k in ("c2","c3","c4","c4sv"): print("fast",k,f[k]["ms_per_step"], f[k]["step_roofline"]["frac"], f[k]["rewrite_kernel_frac_of_hbm_peak"])
print({k:(s[k]["value"], s[k]["wall_s"]) for k in s if k.startswith("e2e")})
PY
head -3 $O/kernel_stats_fast_c*.txt
