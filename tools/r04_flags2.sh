set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04flags2
O=gpurun_out/r04flags2
timeout -k 10 900 python3 -m pytest tests/test_gpu_headline.py tests/test_gpu_edge.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for rw in 1 0; do
python3 bench.py --steps 48 --no-cpu-baseline --ready-words $rw > $O/north_rw$rw.json 2> $O/north_rw$rw.err
python3 bench.py --steps 48 --shard 0/8 --no-cpu-baseline --ready-words $rw > $O/n8_rw$rw.json 2> $O/n8_rw$rw.err
python3 bench.py --steps 48 --shard 0/4 --no-cpu-baseline --ready-words $rw > $O/n4_rw$rw.json 2> $O/n4_rw$rw.err
python3 bench.py --steps 48 --shard 0/2 --no-cpu-baseline --ready-words $rw > $O/n2_rw$rw.json 2> $O/n2_rw$rw.err
done
python3 bench.py --steps 48 --workload cfg3 --no-cpu-baseline > $O/cfg3.json 2> $O/cfg3.err
python3 bench.py --steps 48 --workload cfg5 --shard 0/8 --no-cpu-baseline > $O/cfg5s.json 2> $O/cfg5s.err
python3 bench.py --steps 48 --workload cfg5 --shard 0/8 --no-cpu-baseline --ready-words 0 > $O/cfg5s_rw0.json 2> $O/cfg5s_rw0.err
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/rtz_mul_probe.hip -o /tmp/rtzprobe && timeout -k 5 120 /tmp/rtzprobe > $O/rtz_probe.txt 2>&1
echo done
