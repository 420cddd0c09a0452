set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04stride
O=gpurun_out/r04stride
for m in 1 2; do
AVDSP_FIR_LAUNCH_MODE=$m timeout -k 10 600 python3 -m pytest tests/test_gpu_headline.py -x -q -m gpu -k "overlap or every_channel or host_queue" > $O/pytest_m$m.log 2>&1; echo "mode $m pytest rc=$?"; tail -2 $O/pytest_m$m.log
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_headline.py tests/test_gpu_host_queue.py -x -q -m gpu > $O/pytest_auto.log 2>&1; echo "auto pytest rc=$?"; tail -2 $O/pytest_auto.log
for w in "" "--shard 0/2" "--shard 0/4" "--shard 0/8" "--workload cfg5 --shard 0/8"; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $w 2>/dev/null | grep '^{' >> $O/auto.jsonl
done
python3 bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' >> $O/auto.jsonl
echo done
