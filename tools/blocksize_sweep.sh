# Block-size regime of the chain kernels (round-4 review, item 4): the host's period is the block (linux/avdsp_plugin.c:71-98 hands
# dsp_transfer 64 .. 1024 frames).  GPU box, via gpurun:  bash tools/blocksize_sweep.sh TAG [extra bench.py args]
# -> gpurun_out/TAG_blocksize.jsonl (one bench line per workload and block size), summarised by tools/blocksize_summary.py
set -u
TAG="${1:-r05}"; shift || true
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_blocksize.jsonl
: > "$OUT"
for W in north cfg3 cfg3i cfg4; do
  for B in 64 128 256 512 1024 4096; do
    STEPS=$(( 40960 / B )); [ $STEPS -lt 40 ] && STEPS=40; [ $STEPS -gt 320 ] && STEPS=320
    python3 bench.py --workload $W --block $B --steps $STEPS --warmup 10 --no-cpu-baseline "$@" >> "$OUT" 2>> gpurun_out/${TAG}_blocksize.err || echo "{\"failed\": \"$W $B\"}" >> "$OUT"
    echo "$W B=$B done"
  done
done
python3 tools/blocksize_summary.py "$OUT" > gpurun_out/${TAG}_blocksize.md
