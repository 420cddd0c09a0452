# the round's un-profiled bench lines (GPU box):  bash tools/r04_lines.sh  ->  gpurun_out/r04_bench_lines.jsonl
# (default lines sample every 4th launch of the dominant kernel; the "--profile-stride 1000" lines sample none: what a host that does not
# read the kernel timers gets)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r04_bench_lines.jsonl; : > $L
run() { python3 bench.py --steps 48 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' >> $L; echo "$* rc=$?"; }
python3 bench.py --steps 48 2>/dev/null | grep '^{' >> $L
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' >> $L
run --profile-stride 1000
run --shard 0/2; run --shard 0/4; run --shard 0/8
run --shard 0/2 --profile-stride 1000; run --shard 0/4 --profile-stride 1000; run --shard 0/8 --profile-stride 1000
run --workload cfg2; run --workload cfg3; run --workload cfg3i; run --workload cfg4; run --workload cfg5 --shard 0/8
run --workload cfg2 --profile-stride 1000; run --workload cfg3 --profile-stride 1000; run --workload cfg4 --profile-stride 1000
run --overlap 2
run --ready-words 1; run --ready-words 1 --shard 0/8
run --ready-words 0 --ring-wait 0 --profile-stride 1000; run --ready-words 0 --ring-wait 0 --shard 0/8 --profile-stride 1000; run --ready-words 0 --ring-wait 0 --shard 0/8 --profile-stride 1000
run --ready-words 2 --shard 0/8 --profile-stride 1000; run --ready-words 2 --shard 0/2 --profile-stride 1000; run --ready-words 0 --profile-stride 1000
run --fir-launch 0 --profile-stride 1000; run --fir-launch 0 --shard 0/8 --profile-stride 1000
run --workload cfg4 --fir-split 1 --no-verify
run --workload cfg3 --profile-stride 1
run --fir-impl 4 --profile-stride 1000; run --fir-impl 4 --workload cfg4 --profile-stride 1000; run --fir-impl 4 --shard 0/8 --profile-stride 1000
run --fir-lean 0 --profile-stride 1000; run --fir-lean 1 --shard 0/8 --profile-stride 1000
run --host-buffers
wc -l $L
