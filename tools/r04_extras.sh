# the round's side profiles (GPU box):  bash tools/r04_extras.sh
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/profile_tool.sh r04_lane_formats tools/lane_formats_bench.py --blocks 10 > /dev/null 2>&1
python3 tools/lane_formats_bench.py --lane-hw 0 --blocks 5 2>/dev/null | grep lane_hw > gpurun_out/r04_lane_formats_hw0.txt
bash tools/profile_tool.sh r04_instances tools/instances_bench.py --instances 1 64 1024 4096 > /dev/null 2>&1
bash tools/profile_tool.sh r04_interp tools/extras_bench.py program > /dev/null 2>&1
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/rtz_mul_probe.hip -o /tmp/rtzprobe && timeout -k 5 120 /tmp/rtzprobe > gpurun_out/r04_rtz_mul_probe.txt 2>&1
python3 tools/fir_timeline.py cfg4 --fir-impl 1 --blocks 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_cfg4_timeline.txt
python3 tools/fir_timeline.py cfg4 --fir-impl 1 --fir-split 1 --blocks 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_cfg4_split_timeline.txt
python3 tools/fir_timeline.py north --fir-impl 1 --blocks 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_north_timeline.txt
python3 tools/fir_timeline.py north --fir-impl 4 --blocks 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_north_flow_timeline.txt
python3 tools/fir_timeline.py cfg4 --fir-impl 4 --blocks 200 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_cfg4_flow_timeline.txt
ls gpurun_out | grep r04_
