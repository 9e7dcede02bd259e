#!/bin/bash
export TMPDIR=/tmp
timeout 300 python tools/bench_fit_knobs.py 2>&1 | tail -6
