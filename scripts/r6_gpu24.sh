#!/bin/bash
# round 6, GPU call 24: the round's profile capture (kernel traces, PMC passes) + the default bench line
bash scripts/capture_profiles.sh r06 > gpurun_out/g24_capture.log 2>&1
tail -5 gpurun_out/g24_capture.log
ls gpurun_out | grep r06 | head -40
