# launch rules at short blocks, shards (round 5): default against the FIR-bound regime's settings forced
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --workload north --no-cpu-baseline --no-verify --steps 120 --warmup 10 --profile-stride 1000 "$@" 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read()); c = l['config']
print('%-72s step %8.2f us' % (' '.join(sys.argv[1:]), l['ms_per_step'] * 1e3))" "$@"; }
for B in 256 512 768 1024; do for SH in 0/8 0/4 0/2; do
run --shard $SH --block $B
run --shard $SH --block $B --fir-lean 1 --ready-words 2 --fir-launch 1
run --shard $SH --block $B --ready-words 2 --fir-launch 1
run --shard $SH --block $B --fir-launch 1
done; done
