"""PMC workload: one ViT-B/14 forward of 128 views; DVT_MBLOCK / DVT_NT / DVT_GROUP select the tile order."""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402
from dvt_amd.models import PretrainedViTWrapper  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
L.dvt_tune_set(1, -100 - int(os.environ.get("DVT_MBLOCK", "1")))
L.dvt_tune_set(1, -50 - int(os.environ.get("DVT_NT", "0")))
if "DVT_GROUP" in os.environ:
    L.dvt_tune_set(1, int(os.environ["DVT_GROUP"]))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vit = PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
x = torch.randn(128, 3, 518, 518, device=dev)
out = torch.empty(128, 37, 37, 768, device=dev)
vit.features_nhwc(x, out=out)
torch.cuda.synchronize()
