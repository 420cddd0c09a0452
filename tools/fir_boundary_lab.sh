#!/usr/bin/env bash
# A/B on one box (GPU box, via gpurun): fir_tile's long chunk boundary (round 3: a masked offset and a class look per window sample, a
# clamped taps copy) against the lean one (round 4; fir_tile, LEAN), per workload, un-sampled steps and the FIR kernel with the chip to itself.
#   bash tools/fir_boundary_lab.sh > gpurun_out/r04_fir_boundary_lab.txt
set -uo pipefail
ROOT="$(pwd)"
line() { python3 "$ROOT/bench.py" --no-cpu-baseline --steps 48 "$@" 2>/dev/null | grep '^{' | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
print('ms/step %.4f   FIR launch %.4f   FIR alone %s' % (d['ms_per_step'], r.get('launch_ms',0), ('%.4f' % r['launch_ms_alone']) if r.get('launch_ms_alone') else '-'))"; }
for W in "--workload north" "--workload north --shard 0/2" "--workload north --shard 0/4" "--workload north --shard 0/8" "--workload cfg5 --shard 0/8" "--workload cfg4"; do
  for rep in 1 2; do
    for lean in 0 1; do
      echo -n "$W  fir_lean $lean  un-sampled: "; line $W --fir-lean $lean --profile-stride 1000 | cut -c1-15
      echo -n "$W  fir_lean $lean  stride 4:   "; line $W --fir-lean $lean
    done
  done
done
