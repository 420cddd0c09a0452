set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04inst
O=gpurun_out/r04inst
timeout -k 10 600 python3 -m pytest tests/test_gpu_instances.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout -k 10 300 python3 tools/instances_bench.py > $O/inst_lv6.txt 2>&1; cat $O/inst_lv6.txt | grep -v amdgpu.ids
echo done
