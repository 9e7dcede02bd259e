"""CPU checks of the bf16x3 restatement (oracle/bf16x3.py): the error model the GPU tests of `--fp32_matmul high` rely on."""
import torch

from oracle import bf16x3


def test_split_is_exact_to_2_pow_minus_16():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * torch.exp(4.0 * torch.randn(4096, generator=g))
    hi, lo = bf16x3.split(x)
    assert torch.equal(hi, x.to(torch.bfloat16))
    rec = hi.float() + lo.float()
    assert float(((rec - x).abs() / x.abs()).max()) <= 2.0 ** -16
    # the residual handed to the second rounding is exactly representable: no information is lost before it
    assert torch.equal((x - hi.float()).double(), x.double() - hi.double())


def test_layouts_and_three_term_product():
    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(64, 96, generator=g), torch.randn(48, 96, generator=g) / 96 ** 0.5
    a3, w3 = bf16x3.split3_activation(x), bf16x3.split3_weight(w)
    assert a3.shape == (64, 288) and w3.shape == (48, 288)
    assert torch.equal(a3[:, :96], a3[:, 96:192]) and torch.equal(w3[:, :96], w3[:, 192:])
    (ah, al), (wh, wl) = bf16x3.split(x), bf16x3.split(w)
    three = (ah.double() @ wh.double().T) + (ah.double() @ wl.double().T) + (al.double() @ wh.double().T)
    assert float((a3.double() @ w3.double().T - three).abs().max()) < 1e-12  # the concatenation IS the three-term sum


def test_error_model_between_fp32_and_bf16():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(128, 768, generator=g) * torch.exp(2.0 * torch.randn(128, 1, generator=g))
    w = torch.randn(256, 768, generator=g) / 768 ** 0.5
    b = torch.randn(256, generator=g)
    want = x.double() @ w.double().T + b.double()

    def err(y):
        return float((y.double() - want).norm() / want.norm())
    e3, e32 = err(bf16x3.linear_x3(x, w, b)), err(x @ w.T + b)
    e16 = err(x.bfloat16().float() @ w.bfloat16().float().T + b)
    assert e32 < 1e-6 < e3 < 2e-5 and e16 > 100 * e3, (e32, e3, e16)


def test_attention_restatement_close_to_fp64():
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(200, 64, generator=g) for _ in range(3))
    want = torch.softmax(q.double() @ k.double().T * 0.125, -1) @ v.double()
    got = bf16x3.attention_x3(q, k, v, 0.125)
    assert float((got.double() - want).abs().max() / want.abs().max()) < 5e-5
