set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_host_queue.py tests/test_gpu_edge.py tests/test_gpu_contexts.py -m gpu -x -q > gpurun_out/ov_tests.log 2>&1 || { tail -30 gpurun_out/ov_tests.log; exit 1; }
tail -3 gpurun_out/ov_tests.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config'].get('ready_words'), d['config'].get('ring_wait'))"; }
for rep in 1 2; do
for opt in "" "--ready-words 0 --ring-wait 0"; do
  echo "== default $opt"
  echo -n "shard 0/8: " && timeout -k 10 200 python bench.py --no-cpu-baseline --shard 0/8 --steps 200 --profile-stride 1000 $opt | line
  echo -n "shard 0/4: " && timeout -k 10 200 python bench.py --no-cpu-baseline --shard 0/4 --steps 200 --profile-stride 1000 $opt | line
  echo -n "shard 0/2: " && timeout -k 10 200 python bench.py --no-cpu-baseline --shard 0/2 --steps 100 --profile-stride 1000 $opt | line
  echo -n "north: " && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 48 --profile-stride 1000 $opt | line
  echo -n "north sampled: " && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 48 $opt | line
  echo -n "cfg5 0/8: " && timeout -k 10 200 python bench.py --no-cpu-baseline --workload cfg5 --shard 0/8 --steps 100 --profile-stride 1000 $opt | line
done
done
