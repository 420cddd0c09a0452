# the round's un-profiled bench lines (GPU box): gpurun_out/r04_bench_lines.jsonl
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r04_bench_lines.jsonl; : > $L
run() { python3 bench.py --steps 48 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' >> $L; echo "$* rc=$?"; }
python3 bench.py --steps 48 2>/dev/null | grep '^{' >> $L
run --shard 0/2; run --shard 0/4; run --shard 0/8
run --workload cfg2; run --workload cfg3; run --workload cfg3i; run --workload cfg4; run --workload cfg5 --shard 0/8
run --overlap 2
run --ready-words 1; run --ready-words 1 --shard 0/8
run --workload cfg4 --fir-split 1 --no-verify
run --profile-stride 1 --workload cfg3; run --profile-stride 1000 --workload cfg3
run --host-buffers
wc -l $L
