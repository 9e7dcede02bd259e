#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vit.py -m gpu -q -p no:cacheprovider -k "attention or forward_vs_oracle or layernorm_folded" 2>&1 | tail -3
cat > /tmp/ab.py <<'PY'
import os, sys, time, warnings, torch
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib
from dvt_amd.models import PretrainedViTWrapper
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
x = torch.randn(256, 3, 518, 518, device=dev); out = torch.empty(256, 37, 37, 768, device=dev)
L = _lib.lib()
for tag, v in (("q32", -71), ("q16", -70), ("q32", -71), ("q16", -70)):
    L.dvt_tune_set(1, v)
    vit.features_nhwc(x, out=out); torch.cuda.synchronize()
    _lib.prof_enable(["vit_gemm", "vit_attn"])
    t0 = time.perf_counter(); vit.features_nhwc(x, out=out); torch.cuda.synchronize(); t = time.perf_counter() - t0
    g, a = _lib.prof_collect("vit_gemm"), _lib.prof_collect("vit_attn"); _lib.prof_enable([])
    print(f"{tag}: 256 views {t*1e3:6.1f} ms ({t/256*769*1e3:6.1f} per 769); gemm {g['total_ms']:5.1f} ms {g['work']/g['total_ms']/1e9:6.1f} TF/s; attn {a['total_ms']:5.1f} ms {a['work']/a['total_ms']/1e9:6.1f} TF/s", flush=True)
PY
timeout 300 python /tmp/ab.py 2>&1 | tail -4
