#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/bench_vit_groups.py > gpurun_out/r2e_vit_groups.log 2>&1; tail -10 gpurun_out/r2e_vit_groups.log
