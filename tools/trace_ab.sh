cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in 1 4; do for d in 1 0; do
AVDSP_TIMER_DOUBLES=$d rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_ab_$s$d -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --profile-stride $s > $R/gpurun_out/tr_ab_$s$d.log 2>&1
echo "stride $s doubles $d: $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/tr_ab_$s$d.log)"
rm -rf $R/gpurun_out/tr_ab_$s$d
done; done
