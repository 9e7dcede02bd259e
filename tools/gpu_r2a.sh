#!/bin/bash
# round 2, call A: full GPU test suite (incl. the baseline-shape / chain / cat parity tests) + bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r2a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
grep -E "^\[|passed|failed|FAILED|ERROR|rc=|cosine|cos mean" gpurun_out/r2a_pytest.log | tail -40
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/r2a_bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r2a_bench.log | cut -c1-1500
