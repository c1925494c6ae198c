#!/bin/bash
# final pass: whole GPU suite, the driver's bench command, smoke, then the profile capture
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5q
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
bash scripts/bench_default_check.sh r5q
bash scripts/capture_profiles.sh r05 > $O/capture.log 2>&1; echo "capture rc=$?"
ls gpurun_out | grep r05 | tr '\n' ' '
