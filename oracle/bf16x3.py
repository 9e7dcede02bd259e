"""bf16x3 ("bfloat16_3x") matrix products, restated on the CPU -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Not part of the reference: Jiawei-Yang/Denoising-ViT runs its fp32 matmuls at torch's default float32 matmul precision
("highest").  This file restates what torch.set_float32_matmul_precision("high") PERMITS an fp32 matmul to be (torch docs:
"... or treat each float32 number as the sum of two bfloat16 numbers (approximately 16 bits of mantissa with 10 bits of
the sum of two) ... `bfloat16_3x`"), which the build offers as the opt-in `--fp32_matmul high` of its fp32 extractor
(include/dvt_vit.h: dvt_vit_split3 / dvt_vit_linear_f32x3 / dvt_vit_forward_f32x3).  It pins the arithmetic the HIP path
is tested against: the split, the operand layouts and the three-term product with fp32 accumulation.
"""
import torch


def split(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """x (fp32) -> (hi, lo) bf16 with hi = bf16_rn(x), lo = bf16_rn(x - hi); x - hi is exact in fp32."""
    x = x.float()
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def split3_activation(x: torch.Tensor) -> torch.Tensor:
    """[rows, k] fp32 -> [rows, 3k] bf16 = [hi | hi | lo] (dvt_vit_split3, weights = 0)."""
    hi, lo = split(x)
    return torch.cat([hi, hi, lo], -1)


def split3_weight(w: torch.Tensor) -> torch.Tensor:
    """[n, k] fp32 -> [n, 3k] bf16 = [hi | lo | hi] (dvt_vit_split3, weights = 1)."""
    hi, lo = split(w)
    return torch.cat([hi, lo, hi], -1)


def linear_x3(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None = None) -> torch.Tensor:
    """y = x . w^T + b as ONE product over the K-concatenated splits, accumulated in fp32: a_hi w_hi + a_hi w_lo + a_lo w_hi.
    (bf16 x bf16 products are exact in fp32; only the accumulation order differs from the MFMA's.)"""
    y = split3_activation(x).float() @ split3_weight(w).float().T
    return y if b is None else y + b.float()


def attention_x3(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(scale q k^T) v for one head with both products through three split terms, softmax in fp32
    (dvt_vit_attention_x3; the HIP kernel streams the keys with an online softmax, which is the same function)."""
    qh, ql = split(q * scale)   # scale = 2^-3 for head_dim 64: exact
    kh, kl = split(k)
    s = qh.float() @ kl.float().T + ql.float() @ kh.float().T + qh.float() @ kh.float().T
    p = torch.exp(s - s.max(-1, keepdim=True).values)  # unnormalised, as the kernel holds it (its running max may lag the
    ph, pl = split(p)                                   # true one by up to 8, which moves the split's rounding, not its size)
    vh, vl = split(v)
    o = ph.float() @ vl.float() + pl.float() @ vh.float() + ph.float() @ vh.float()
    return o / p.sum(-1, keepdim=True)
