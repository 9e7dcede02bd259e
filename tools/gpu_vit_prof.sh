#!/bin/bash
# per-kernel timing of the extractor alone (256 views), GEMM variant given as $1
export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/vit_only.py <<PY
import os, sys, warnings, torch
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib
from dvt_amd.models import PretrainedViTWrapper
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14)
x = torch.randn(256, 3, 518, 518, device=dev)
out = torch.empty(256, 37, 37, 768, device=dev)
_lib.lib().dvt_tune_set(1, ${1:-0})
for _ in range(3):
    vit.features_nhwc(x, out=out)
torch.cuda.synchronize()
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_vit -o vit -- python /tmp/vit_only.py > $GRAFT_REPO_ROOT/gpurun_out/prof_vit.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_vit -name '*.db' | head -1) > gpurun_out/prof_vit_stats_v${1:-0}.txt
rm -rf gpurun_out/prof_vit
cat gpurun_out/prof_vit_stats_v${1:-0}.txt | cut -c1-150
