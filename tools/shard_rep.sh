cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for sh in 0/8 0/4; do for st in 4 1000; do
python3 bench.py --steps 48 --no-cpu-baseline --shard $sh --profile-stride $st 2>/dev/null | python3 -c "
import sys,json; l=json.loads(sys.stdin.read()); print('shard $sh stride $st rep $rep: step %.2f us  fir %.1f bq %.1f' % (l['ms_per_step']*1e3, l['kernels_ms']['fir']*1e3, l['kernels_ms']['biquad']*1e3))"
done; done; done
