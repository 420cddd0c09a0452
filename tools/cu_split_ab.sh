# Experiment (round 5): what would a shard's step be if the next blocks' cascades had CUs of their own and the FIR did not
# need a second round of waves on the rest?  The cascades' stream on 16 / 32 CUs (CU mask), the FIRs on a library stream on the others.
#   shard 0/8 (512 chains = 2048 FIR waves): more than 240 / 224 CUs hold at once -> the second round round 3 measured;
#   shard 0/9 (456 chains = 1824 waves) and 0/10 (410 chains = 1640 waves): ONE round on 240 / 224 CUs -> the upper bound of what FIR work
#   handed out in fine units (a persistent grid with parked accumulators) could reach, without building it.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r05}_cu_split_ab.txt; : > "$OUT"
run() { python3 bench.py --no-cpu-baseline --steps 96 --warmup 10 --profile-stride 1000 "$@" 2>>gpurun_out/cu_split.err | python3 -c "
import sys, json
l = json.loads(sys.stdin.read()); c = l['config']
print('%-46s ch %4d  step %7.2f us  %7.1f Msamples/s   fir alone %.1f us  cascade alone %.1f us  side_by_side %s remade %s' % (' '.join(sys.argv[1:]), c['channels_per_gpu'], l['ms_per_step'] * 1e3, l['value'], (l['roofline'] or {}).get('launch_ms_alone', 0) * 1e3, l['kernels_ms']['biquad'] * 1e3, c['side_by_side'], c['streams_remade']))" "$@" >> "$OUT"; }
for SH in 0/8 0/9 0/10 0/4; do
  run --shard $SH
  run --shard $SH --cu-split 16
  run --shard $SH --cu-split 32
done
run --cu-split 16
cat "$OUT"
