#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python tools/bench_grid.py > gpurun_out/r2e_grid.log 2>&1; tail -10 gpurun_out/r2e_grid.log
