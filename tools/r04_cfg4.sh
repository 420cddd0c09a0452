set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04cfg4
O=gpurun_out/r04cfg4
timeout -k 10 900 python3 -m pytest tests/test_gpu_fir_tile.py -x -q -m gpu -k "split or fir_only" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
python3 bench.py --steps 48 --no-cpu-baseline --workload cfg4 > $O/cfg4.json 2> $O/cfg4.err
python3 bench.py --steps 48 --no-cpu-baseline --workload cfg4 --fir-split 1 --no-verify > $O/cfg4_split.json 2> $O/cfg4_split.err
python3 bench.py --steps 48 --no-cpu-baseline --workload cfg4 --fir-impl 3 > $O/cfg4_stream.json 2> $O/cfg4_stream.err
python3 bench.py --steps 48 --no-cpu-baseline --workload cfg4 --fir-rows 1 > $O/cfg4_rows1.json 2> $O/cfg4_rows1.err
echo done
