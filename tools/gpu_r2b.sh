#!/bin/bash
# round 2, call B: fused row kernel -- parity (fused vs layer-by-layer, baseline shapes vs oracle), timing, bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_parity_full.py -m gpu -q -s -p no:cacheprovider -k "fused or baseline or batched or full_size" > gpurun_out/r2b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b_pytest.log
grep -E "^\[|passed|failed|FAILED|ERROR|rc=|fused vs|Error|error" gpurun_out/r2b_pytest.log | cut -c1-400 | tail -30
timeout 600 python tools/bench_fit_modes.py > gpurun_out/r2b_fit_modes.log 2>&1; cat gpurun_out/r2b_fit_modes.log | tail -5
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r2b_bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r2b_bench.log | cut -c1-400
