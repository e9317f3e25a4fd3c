cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ah
timeout -s KILL 900 python -m pytest tests/test_gpu_bench_order.py tests/test_gpu_sampler.py tests/test_gpu_ahead.py tests/test_gpu_stream_rebase.py -q -x > gpurun_out/ah/pytest_part.txt 2>&1; echo "rc=$?"
for i in 1 2 3; do
echo "== maps deferred"; python3 mutation-simulator_amd/tools/compat_steps.py c2 20 3e9 | head -2
echo "== one pass"; MSIM_NO_MAPS_DEFER=1 python3 mutation-simulator_amd/tools/compat_steps.py c2 20 3e9 | head -2
done
grep "passed\|failed" gpurun_out/ah/pytest_part.txt | tail -2
