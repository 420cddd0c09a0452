set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04lane
O=gpurun_out/r04lane
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sweeps.py -x -q -m gpu -k "float_accumulator or int_sample or golden or chain_shapes" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
python3 tools/lane_formats_bench.py --lane-hw 1 > $O/lane_hw1.txt 2>&1
cat $O/lane_hw1.txt
echo done
