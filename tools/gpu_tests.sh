#!/bin/bash
# tools/gpu_tests.sh [pytest args...]: the GPU suite in ONE process on the GPU box, with the complete log kept under
# gpurun_out/ (gpurun returns only the tail of stdout) and every mismatching array saved by tests/helpers.assert_same
# under gpurun_out/test_failures/ — a failure must be explainable from the one run that showed it.
mkdir -p gpurun_out/test_failures
log=gpurun_out/pytest_gpu_$(date +%H%M%S).log
python3 -m pytest tests -m gpu -q -x -rfE --durations=15 "$@" > "$log" 2>&1
rc=$?
tail -n 25 "$log"
echo "pytest rc=$rc; full log in $log"
exit $rc
