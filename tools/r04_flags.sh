set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04flags
O=gpurun_out/r04flags
timeout -k 10 600 python3 -m pytest tests/test_gpu_headline.py -x -q -m gpu -k "overlap or every_channel" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for rw in 1 0; do
python3 bench.py --steps 50 --no-cpu-baseline --ready-words $rw > $O/north_rw$rw.json 2> $O/north_rw$rw.err
python3 bench.py --steps 50 --shard 0/8 --no-cpu-baseline --ready-words $rw > $O/n8_rw$rw.json 2> $O/n8_rw$rw.err
python3 bench.py --steps 50 --shard 0/4 --no-cpu-baseline --ready-words $rw > $O/n4_rw$rw.json 2> $O/n4_rw$rw.err
python3 bench.py --steps 50 --shard 0/2 --no-cpu-baseline --ready-words $rw > $O/n2_rw$rw.json 2> $O/n2_rw$rw.err
done
for st in 1 4 1000; do
python3 bench.py --steps 50 --workload cfg3 --no-cpu-baseline --profile-stride $st > $O/cfg3_st$st.json 2> $O/cfg3_st$st.err
python3 bench.py --steps 50 --workload cfg4 --no-cpu-baseline --profile-stride $st > $O/cfg4_st$st.json 2> $O/cfg4_st$st.err
python3 bench.py --steps 50 --no-cpu-baseline --profile-stride $st > $O/north_st$st.json 2> $O/north_st$st.err
done
python3 bench.py --steps 50 --no-cpu-baseline --overlap 2 > $O/north_ov2.json 2> $O/north_ov2.err
python3 bench.py --steps 50 --no-cpu-baseline --overlap 2 --ready-words 0 > $O/north_ov2_rw0.json 2> $O/north_ov2_rw0.err
echo done
