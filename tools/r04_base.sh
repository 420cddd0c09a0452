set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04base
O=gpurun_out/r04base
python3 bench.py --steps 50 > $O/north.json 2> $O/north.err
python3 bench.py --steps 50 --shard 0/8 --no-cpu-baseline > $O/n8.json 2> $O/n8.err
python3 bench.py --steps 50 --shard 0/4 --no-cpu-baseline > $O/n4.json 2> $O/n4.err
python3 bench.py --steps 50 --shard 0/2 --no-cpu-baseline > $O/n2.json 2> $O/n2.err
python3 bench.py --steps 50 --workload cfg3 --no-cpu-baseline > $O/cfg3.json 2> $O/cfg3.err
python3 bench.py --steps 50 --workload cfg4 --no-cpu-baseline > $O/cfg4.json 2> $O/cfg4.err
python3 bench.py --steps 50 --workload cfg2 --no-cpu-baseline > $O/cfg2.json 2> $O/cfg2.err
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr8 -- python3 $R/bench.py --shard 0/8 --no-cpu-baseline --steps 50 > $R/$O/tr8.log 2>&1
python3 $R/tools/step_gaps.py /tmp/tr8 > $R/$O/gaps8.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -- python3 $R/bench.py --no-cpu-baseline --steps 50 > $R/$O/tr1.log 2>&1
python3 $R/tools/step_gaps.py /tmp/tr1 > $R/$O/gaps1.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr3 -- python3 $R/bench.py --workload cfg3 --no-cpu-baseline --steps 50 > $R/$O/tr3.log 2>&1
python3 $R/tools/step_gaps.py /tmp/tr3 > $R/$O/gaps3.txt 2>&1
echo done
