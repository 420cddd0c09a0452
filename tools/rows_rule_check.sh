cd $GRAFT_REPO_ROOT
run() { python3 bench.py --custom "$1" --no-cpu-baseline --steps 60 --warmup 6 --profile-stride 1000 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read()); r = l['roofline'] or {}
print('%-24s step %9.2f us  fir alone %8.1f us  frac %.3f' % ('$1', l['ms_per_step'] * 1e3, r.get('launch_ms', 0) * 1e3, r.get('frac', 0)))"; }
for C in 700 1000 1027 1500 2500 3000 3500 5000 6000; do run 6,$C,16,4096,1024; done
for W in north cfg5; do python3 bench.py --workload $W --shard 0/8 --no-cpu-baseline --steps 48 --profile-stride 1000 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('$W 0/8', round(l['ms_per_step']*1e3,2), bool(l['verified']))"; done
python3 bench.py --no-cpu-baseline --steps 48 --profile-stride 1000 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('north', round(l['ms_per_step']*1e3,2), bool(l['verified']))"
