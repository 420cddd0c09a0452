set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04lean
O=gpurun_out/r04lean
timeout -k 10 1000 python3 -m pytest tests/test_gpu_fir_tile.py tests/test_gpu_edge.py tests/test_gpu_headline.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
: > $O/lines.jsonl
for w in "" "--shard 0/2" "--shard 0/4" "--shard 0/8" "--workload cfg5 --shard 0/8" "--workload cfg4"; do
python3 bench.py --steps 48 --no-cpu-baseline --profile-stride 1000 $w 2>/dev/null | grep '^{' >> $O/lines.jsonl
python3 bench.py --steps 48 --no-cpu-baseline $w 2>/dev/null | grep '^{' >> $O/lines.jsonl
done
echo done
