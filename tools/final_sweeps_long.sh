# a longer pass of every sweep with fresh seeds (GPU box, ~15 minutes):  bash tools/final_sweeps_long.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_final_sweeps_long2.txt; : > $O
run() { echo "== $*" >> $O; timeout -k 10 $1 ${@:2} 2>&1 | tail -n 2 >> $O; echo "rc=$?" >> $O; }
AVDSP_SWEEP_ALL_FORMATS=1 run 300 python tests/dev/gpu_chain_sweep.py 22000 22500
run 200 python tests/dev/gpu_fuzz_sweep.py 22000 22600
AVDSP_FUZZ_NAN_HEAVY=1 run 150 python tests/dev/gpu_fuzz_sweep.py 23000 23300
AVDSP_FUZZ_WIDE=1 run 150 python tests/dev/gpu_fuzz_sweep.py 24000 24200
run 200 python tests/dev/gpu_wave_sweep.py 22000 22400 300 all
AVDSP_SWEEP_OVERLAP=1 run 150 python tests/dev/gpu_wave_sweep.py 23000 23200 300 all
run 200 python tests/dev/gpu_strand_sweep.py 22000 22500
run 200 python tests/dev/gpu_overlap_sweep.py 22000 22150
run 200 python tests/dev/gpu_instance_sweep.py 22000 22300
run 150 python tests/dev/gpu_wide_blocks_sweep.py 22000 22040
AVDSP_SWEEP_OPTIONS="fir_rows=2" run 150 python tests/dev/gpu_chain_sweep.py 25000 25200
AVDSP_SWEEP_OPTIONS="biquad_impl=2" run 150 python tests/dev/gpu_chain_sweep.py 25200 25400
AVDSP_SWEEP_OPTIONS="group_fanout=0" run 150 python tests/dev/gpu_chain_sweep.py 25400 25600
cat $O
