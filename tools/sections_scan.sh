cd $GRAFT_REPO_ROOT
run() { python3 bench.py --custom "$1" --no-cpu-baseline --steps 40 --warmup 4 --profile-stride 1000 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read())
f,C,S,T,B = [int(v) for v in '$1'.split(',')]
print('%-24s step %9.2f us  %8.2f Gsamples/s  %6.1f ns per (section x frame x 1000 chains)' % ('$1', l['ms_per_step'] * 1e3, l['value'] / 1e3, l['ms_per_step']*1e6/(S*B*C/1000.0)))"; }
for S in 8 16 17 24 32 33 48 64 65 100 128 200; do run 6,4096,$S,0,1024; done
for S in 16 17 32 65 128; do run 2,4096,$S,0,1024; done
run 4,4096,65,0,1024
for C in 100 1000 2000 3000 5000 10000; do run 6,$C,16,0,1024; done
