cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/pageable_copy_probe tools/pageable_copy_probe.hip || exit 2
AMD_LOG_LEVEL=4 /tmp/pageable_copy_probe > /tmp/probe.log 2>&1
# per copy: the runtime's own lines that say how it moved the bytes (addresses and time stamps cut off)
awk '/^===/ {print; next} /[Cc]opy|[Pp]in|[Ss]tag|Blit|SDMA|sdma/ {sub(/^[^\]]*\] /, ""); print "    " substr($0, 1, 150)}' /tmp/probe.log | uniq -c > gpurun_out/r05_pageable_copy_probe.txt
head -150 gpurun_out/r05_pageable_copy_probe.txt
