set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/traffic.json
TRAFFIC_KEY="north" bash tools/profile_gpu.sh r03_north --workload north
TRAFFIC_KEY="cfg3" bash tools/profile_gpu.sh r03_cfg3 --workload cfg3
TRAFFIC_KEY="cfg3i" bash tools/profile_gpu.sh r03_cfg3i --workload cfg3i
TRAFFIC_KEY="cfg4" bash tools/profile_gpu.sh r03_cfg4 --workload cfg4
TRAFFIC_KEY="north shard 0/8" bash tools/profile_gpu.sh r03_north8 --workload north --shard 0/8
TRAFFIC_KEY="cfg5 shard 0/8" bash tools/profile_gpu.sh r03_cfg5s --workload cfg5 --shard 0/8
ls gpurun_out | grep r03
