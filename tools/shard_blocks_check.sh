cd $GRAFT_REPO_ROOT
run() { python3 bench.py --workload north --no-cpu-baseline --steps 120 --warmup 10 --profile-stride 1000 "$@" 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read()); c = l['config']
print('%-40s step %8.2f us  verified %s' % (' '.join(sys.argv[1:]), l['ms_per_step'] * 1e3, bool(l['verified'])))" "$@"; }
for B in 256 512 768 1024; do for SH in 0/8 0/4; do run --shard $SH --block $B; done; done
run --workload cfg5 --shard 0/8 --block 256; run --workload cfg5 --shard 0/8 --block 512; run --workload cfg5 --shard 0/8
run --workload cfg4 --block 256; run --workload cfg4
