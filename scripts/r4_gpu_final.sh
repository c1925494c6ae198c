#!/bin/bash
# final verification of the round: full GPU suite, profile capture, default bench
ulimit -c 0
mkdir -p gpurun_out/r4f
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4f/gputests.txt 2>&1; tail -4 gpurun_out/r4f/gputests.txt
timeout 1500 bash scripts/capture_profiles.sh r04 > gpurun_out/r4f/capture.log 2>&1; tail -3 gpurun_out/r4f/capture.log
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; tail -2 gpurun_out/r4f/bench.err; cut -c1-600 gpurun_out/r4f/bench.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
