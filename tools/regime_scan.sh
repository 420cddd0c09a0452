# A scan over program shapes for launch arrangements that do not fit (round 5: two were found at short blocks by looking).  GPU box:
#   bash tools/regime_scan.sh > gpurun_out/regime_scan.txt
# prints step, FIR launch alone and its fraction of the FP64 MFMA peak; anything far below its neighbours is worth a look
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --custom "$1" --no-cpu-baseline --steps 60 --warmup 6 --profile-stride 1000 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read()); c = l['config']; r = l['roofline'] or {}
print('%-24s step %9.2f us  %8.2f Gsamples/s   %-11s alone %8.1f us  frac %.3f   cascade %6.1f us' % ('$1', l['ms_per_step'] * 1e3, l['value'] / 1e3, r.get('kernel', '-'), r.get('launch_ms', 0) * 1e3, r.get('frac', 0), l['kernels_ms']['biquad'] * 1e3))"; }
for B in 256 1024; do
for T in 512 2048 4096; do
for C in 128 256 512 1024 2048 4096 8192; do
run 6,$C,8,$T,$B
done; done; done
for C in 64 300 1000 3000 10000; do run 6,$C,16,4096,1024; run 6,$C,0,1024,1024; run 4,$C,4,300,512; done
