cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_final_sweeps.txt; : > $O
run() { echo "== $*" >> $O; timeout -k 10 $1 ${@:2} 2>&1 | tail -n 3 >> $O; echo "rc=$?" >> $O; }
run 240 python tests/dev/gpu_fuzz_sweep.py 9000 9400
run 240 python tests/dev/gpu_wave_sweep.py 9000 9300 300 all
run 200 python tests/dev/gpu_strand_sweep.py 9000 9300
run 200 python tests/dev/gpu_overlap_sweep.py 9000 9080
run 200 python tests/dev/gpu_wide_blocks_sweep.py 9000 9030
cat $O
