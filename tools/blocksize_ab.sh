# The launch arrangement's choices at short blocks (round 5): lean FIR boundary, ready words, FIR launch mode -- each forced both ways
# on the north-star program at 256 and 512 frames (and 1024 for reference).  GPU box:  bash tools/blocksize_ab.sh TAG
set -u
TAG="${1:-r05}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_blocksize_ab.txt
: > "$OUT"
run() { python3 bench.py --workload north --no-cpu-baseline --no-verify --steps 120 --warmup 10 --profile-stride 1000 "$@" 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read()); c = l['config']
print('B %4d  %-40s step %8.2f us   (lean %s ready_mode %s launch %s)' % (c['block'], ' '.join(sys.argv[1:]), l['ms_per_step'] * 1e3, c['fir_lean'], c['ready_mode'], c['fir_launch']))" "$@" >> "$OUT"; }
for B in 256 512 1024; do
  run --block $B
  run --block $B --fir-lean 0
  run --block $B --fir-lean 1
  run --block $B --ready-words 0
  run --block $B --ready-words 2
  run --block $B --fir-launch 1
  run --block $B --fir-launch 2
  run --block $B --fir-lean 1 --ready-words 2 --fir-launch 1
  run --block $B --fir-lean 0 --ready-words 0 --fir-launch 2
done
cat "$OUT"
