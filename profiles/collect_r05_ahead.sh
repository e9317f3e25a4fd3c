#!/bin/bash
# Round-5 evidence for the SNP sampler's anchored windows (ranks of a sharded step): A/B of tools/compat_steps.py with and without
# them, kernel timelines of a rank that owns every eighth contig, and the driver's bench line on the final binary.
# Same rules as collect_r05.sh; summaries land in gpurun_out/r5ev/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ev; mkdir -p $O
T="timeout -s KILL 300"
S="python3 mutation-simulator_amd/tools/compat_steps.py c2 20"
( for i in 1 2 3; do
    echo "# windows anchored ahead of the chain on sharded ranks (default)"; $T $S
    echo "# MSIM_NO_AHEAD=1: every sample on the chain"; MSIM_NO_AHEAD=1 $T $S
  done ) > $O/sharded_rank_steps.txt 2>&1
for v in ahead chain; do
  [ $v = chain ] && export MSIM_NO_AHEAD=1
  $T rocprofv3 --kernel-trace --stats -d $O/ks_$v -o ks -- python3 mutation-simulator_amd/tools/compat_steps.py c2 3 3e9 8 > $O/ks_$v.log 2>&1
  python3 profiles/summarize_rocprof.py timeline $O/ks_$v/ks_results.db -2 330 2 > $O/timeline_c2_rank_owning_3_of_24_$v.txt 2>&1
  rm -rf $O/ks_$v
done
unset MSIM_NO_AHEAD
$T python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r5ev/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"].get("matches_gpu"))
print(d.get("amdahl_ceiling"))
s=d["secondary"]
for k in ("c3","c4","c4sv"): print(k, s[k]["value"], s[k]["ms_per_step"], s[k]["roofline"]["frac"], s[k].get("amdahl_ceiling"))
PY
cat $O/sharded_rank_steps.txt
