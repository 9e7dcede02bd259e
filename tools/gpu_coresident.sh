#!/bin/bash
# does letting the fit's LDS-using kernels co-reside with the ViT GEMM workgroups help the pipeline?
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fit.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
run() { echo "== tune='$1' $2"; timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-probes --tune "$1" $2 > gpurun_out/b.log 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/b.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print("   images/s", round(d["value"],3), "ms/img", round(d["ms_per_step"],1), "serial ext/fit", round(d["config"]["t_extract_s_serial"],3), round(d["config"]["t_fit_s_serial"],3))
else:
    print(open('gpurun_out/b.log').read()[-600:])
PY
}
run ""                    ""
run "5=16"                ""
run "5=16,2=-256"         ""
run "5=32,2=-256"         ""
run "5=16,2=-256,1=1"     ""
run "5=16,2=-256,1=3"     ""
run "5=16,2=0"            ""
run "5=16,2=-256"         "--pipeline-depth 1"
