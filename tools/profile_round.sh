# The round's bench.py profiles in one call (GPU box, via gpurun):  bash tools/profile_round.sh r04
set -u
R="${1:-r05}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/traffic.json
TRAFFIC_KEY="north" bash tools/profile_gpu.sh ${R}_north --workload north
TRAFFIC_KEY="cfg2" bash tools/profile_gpu.sh ${R}_cfg2 --workload cfg2
TRAFFIC_KEY="cfg3" bash tools/profile_gpu.sh ${R}_cfg3 --workload cfg3
TRAFFIC_KEY="cfg3i" bash tools/profile_gpu.sh ${R}_cfg3i --workload cfg3i
TRAFFIC_KEY="cfg4" bash tools/profile_gpu.sh ${R}_cfg4 --workload cfg4
TRAFFIC_KEY="north shard 0/8" bash tools/profile_gpu.sh ${R}_north8 --workload north --shard 0/8
TRAFFIC_KEY="cfg5 shard 0/8" bash tools/profile_gpu.sh ${R}_cfg5s --workload cfg5 --shard 0/8
ls gpurun_out | grep ${R}
TRAFFIC_KEY="north block 256" bash tools/profile_gpu.sh ${R}_north_b256 --workload north --block 256
ls gpurun_out | grep ${R}
