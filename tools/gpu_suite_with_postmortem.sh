# The whole -m gpu suite; if the process dies of a GPU memory fault the runtime leaves gpucore.<pid> in the working directory -- rocgdb then says
# WHAT faulted (a wave of which kernel at which pc, or no wave at all: a copy engine).   bash tools/gpu_suite_with_postmortem.sh [log name]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LOG=gpurun_out/${1:-gpu_suite}.log
rm -f gpucore.*
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $LOG 2>&1
rc=$?
echo rc=$rc >> $LOG
for c in gpucore.*; do
    [ -f "$c" ] || continue
    ls -la $c >> $LOG
    timeout -k 5 120 /opt/rocm/bin/rocgdb --batch -ex "info agents" -ex "info queues" -ex "info dispatches" -ex "info threads" -ex "thread apply all bt 6" \
        -ex "thread apply all x/6i \$pc" $(readlink -f $(which python3)) $c > gpurun_out/${1:-gpu_suite}_gpucore.txt 2>&1
    tail -60 gpurun_out/${1:-gpu_suite}_gpucore.txt
done
tail -4 $LOG
exit $rc
