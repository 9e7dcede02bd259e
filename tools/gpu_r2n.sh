#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/bench_stage1_files.py --images 16 > gpurun_out/r2n_files.log 2>&1; echo rc=$?; grep -E "^files|images in|Error|error" gpurun_out/r2n_files.log | tail -5
