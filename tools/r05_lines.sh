# the round's un-profiled bench lines (GPU box):  bash tools/r05_lines.sh  ->  gpurun_out/r05_bench_lines.jsonl
# (default lines: every launch of a dominant kernel of >= 0.2 ms carries its stamps, every fourth of a shorter one; "--profile-stride 1000": none)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r05_bench_lines.jsonl; : > $L
run() { python3 bench.py --steps 48 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' >> $L; echo "$* rc=$?"; }
python3 bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' >> $L
python3 bench.py --steps 48 2>/dev/null | grep '^{' >> $L
run --profile-stride 4; run --profile-stride 1000
run --shard 0/2; run --shard 0/4; run --shard 0/8
run --shard 0/2 --profile-stride 1000; run --shard 0/4 --profile-stride 1000; run --shard 0/8 --profile-stride 1000
run --workload cfg2; run --workload cfg3; run --workload cfg3i; run --workload cfg4; run --workload cfg5 --shard 0/8
run --workload cfg2 --profile-stride 1000; run --workload cfg3 --profile-stride 1000; run --workload cfg3i --profile-stride 1000; run --workload cfg4 --profile-stride 1000
run --block 256; run --block 512; run --block 256 --profile-stride 1000; run --block 512 --profile-stride 1000
run --workload cfg3 --block 256 --profile-stride 1000; run --workload cfg3i --block 256 --profile-stride 1000
run --overlap 2
run --ready-words 0 --profile-stride 1000; run --ready-words 0 --overlap 0 --profile-stride 1000
run --host-buffers
wc -l $L
