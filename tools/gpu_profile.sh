#!/bin/bash
# rocprofv3 kernel traces (no event probes): fit alone, full bench serial, full bench pipelined
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cat > /tmp/fit_only.py <<'PY'
import os, sys, time, numpy as np, torch
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.fit import FitEngine, FitSettings
dev = torch.device("cuda:0"); n_rows = 769 * 1369
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, 768, device=dev, generator=g); xy = torch.rand(n_rows, 2, device=dev, generator=g)
eng = FitEngine(FitSettings(num_iters=1000, warmup_iters=100), n_rows, dev)
np.random.seed(0)
for rep in range(2):
    eng.reset(g); torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.fit(feat, xy, None, log_every=1000); torch.cuda.synchronize()
    print("fit us/step", (time.perf_counter() - t0) * 1e3)
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fit -o fit -- python /tmp/fit_only.py > $GRAFT_REPO_ROOT/gpurun_out/prof_fit.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_serial -o serial -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probes --pipeline-depth 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pipe -o pipe -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-probes > $GRAFT_REPO_ROOT/gpurun_out/prof_pipe.log 2>&1
cd $GRAFT_REPO_ROOT
for n in fit serial pipe; do python tools/rocpd_stats.py $(find gpurun_out/prof_$n -name '*.db' | head -1) > gpurun_out/prof_${n}_stats.txt; tail -1 gpurun_out/prof_$n.log | cut -c1-160; done
python tools/gap_attrib.py gpurun_out/prof_pipe > gpurun_out/prof_pipe_gaps.txt 2>&1
rm -rf gpurun_out/prof_fit gpurun_out/prof_serial gpurun_out/prof_pipe
timeout 600 python bench.py --steps 8 --warmup 1 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-200
