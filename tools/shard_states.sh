# The two states of a small shard's step, from rocprofv3 kernel traces of bench.py --shard 0/8:
# the cascades' stream waiting for the FIR three blocks back ("ring_wait" 0) against the host doing it (1, the default).
#   bash tools/shard_states.sh        (GPU box; writes gpurun_out/r04_shard_states.txt)
set -u
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; OUT="$R/gpurun_out/r04_shard_states.txt"; : > "$OUT"
cd /tmp && export TMPDIR=/tmp
for rw in 0 1; do
  for i in 1 2 3; do
    rm -rf /tmp/trs
    rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -- python3 "$R/bench.py" --shard 0/8 --steps 200 --profile-stride 1000 --no-cpu-baseline --ready-words 0 --ring-wait $rw > /tmp/trs.log 2>&1
    echo "== ring_wait $rw, run $i: $(python3 -c "import json; d=json.loads([l for l in open('/tmp/trs.log') if l.startswith('{')][-1]); print('ms_per_step', round(d['ms_per_step'], 4))")" >> "$OUT"
    python3 "$R/tools/step_gaps.py" /tmp/trs 2>&1 | head -6 >> "$OUT"
  done
done
cat "$OUT"
