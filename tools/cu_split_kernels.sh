cd $GRAFT_REPO_ROOT
for a in "--cu-split -16 --own-stream" "--cu-split -32 --own-stream" "--cu-split -64 --own-stream" "--cu-split 8 --own-stream" "--cu-split 24 --own-stream"; do
for sh in 0/8 0/4; do
python3 bench.py --no-cpu-baseline --no-verify --steps 96 --warmup 10 --profile-stride 1000 --shard $sh $a 2>/dev/null | python3 -c "
import sys,json; l=json.loads(sys.stdin.read()); print('shard $sh $a', 'step %.1f us' % (l['ms_per_step']*1e3))"
done; done
