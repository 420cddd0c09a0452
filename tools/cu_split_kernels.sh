cd $GRAFT_REPO_ROOT
for a in "--own-stream" "--cu-split -32 --own-stream" "--cu-split -64 --own-stream" "--cu-split -128 --own-stream"; do
python3 bench.py --no-cpu-baseline --no-verify --steps 48 --warmup 10 --profile-stride 1000 $a 2>/dev/null | python3 -c "
import sys,json; l=json.loads(sys.stdin.read()); print('north $a', 'step %.1f us' % (l['ms_per_step']*1e3))"
done
