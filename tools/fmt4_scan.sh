cd $GRAFT_REPO_ROOT
run() { python3 bench.py --custom "$1" --no-cpu-baseline --steps 40 --warmup 4 --profile-stride 1000 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read())
print('%-24s step %9.2f us' % ('$1', l['ms_per_step'] * 1e3))"; }
for c in 4,4096,16,0,1024 4,4096,8,0,1024 6,4096,16,0,1024 2,4096,16,0,1024 4,4096,16,512,1024 6,4096,16,512,1024; do run $c; done
