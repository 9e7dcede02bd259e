"""debug: NaN at batch 1 with row_pad 32"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.vit import HipViT, random_state_dict
DEV = "cuda"
sd = random_state_dict(768, 2, 14, 1370, seed=768, well_conditioned=True)
x = torch.randn(2, 3, 518, 518, generator=torch.Generator().manual_seed(1)).to(DEV)
vit = HipViT(sd, 14, 14, (518, 518), DEV)
cfg = vit.cfg
print("s_pad", cfg.s_pad)
def up(b): return (b + 255) // 256 * 256
def carve(batch):
    T = (batch * cfg.s_pad + 255) // 256 * 256
    D, F, KP = cfg.dim, cfg.mlp_dim, cfg.k_patch
    sizes = [("x", T * D * 4, torch.float32), ("xn", T * D * 2, torch.bfloat16), ("qk", (T + 128) * 2 * D * 2, torch.bfloat16),
             ("vt", (batch + 1) * cfg.s_pad * D * 2, torch.bfloat16), ("hid", T * F * 2, torch.bfloat16), ("col", T * KP * 2, torch.bfloat16),
             ("xb", T * D * 2, torch.bfloat16), ("st_part", (D // 64) * T * 8, torch.float32), ("stats", T * 8, torch.float32)]
    o, out = 0, {}
    for n, b, dt in sizes:
        out[n] = (o, b, dt)
        o += up(b)
    return T, out
for nb in (0, 1, 2):
    for batch in (2, 1):
        got = vit.forward_features(x[:batch], n_blocks=nb, max_batch=batch)
        torch.cuda.synchronize()
        T, lay = carve(batch)
        msg = []
        for n, (o, b, dt) in lay.items():
            t = vit._ws[o:o + b].view(dt).float()
            bad = int((~torch.isfinite(t)).sum())
            if bad:
                rows = (~torch.isfinite(t.view(-1, t.numel() // (T + (128 if n == "qk" else 0)) if n not in ("vt",) else t.numel()))).any(1).nonzero().flatten() if n != "vt" else None
                msg.append(f"{n}: {bad} non-finite" + (f" rows {rows[:6].tolist()}..{rows[-3:].tolist()}" if rows is not None and len(rows) else ""))
        print(f"n_blocks {nb} batch {batch}: out non-finite {int((~torch.isfinite(got)).sum())}; " + "; ".join(msg))
