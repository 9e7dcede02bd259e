"""CPU statements of arithmetic the bf16 extractor relies on: two pieces of its fc1 epilogue (csrc/dvt_vit.hip, gelu_erf /
gelu_erf_pair); the kernels themselves are held against the oracle by tests/test_gpu_vit.py.

1. erf-GELU through Abramowitz-Stegun 7.1.28 with the powers of 1/sqrt(2) folded into the coefficients:
   GELU(x) = max(x, 0) - 0.5 |x| (1 + c1 |x| + ... + c6 |x|^6)^-16  (timm Mlp's nn.GELU(): reference vit_wrapper.py:105-120
   builds the timm model whose blocks call it).
2. max(x, 0) = 0.5 x + 0.5 |x| in fp32 arithmetic -- what lets the packed-fp32 version replace two v_max_f32 per element by
   one half of a v_pk_fma_f32 without changing a bit.
3. (round 6, further down) the tile map's multiply-shift division and the log2-domain attention loop.
"""
import numpy as np
from scipy.special import erfc

# the literals of gelu_erf (highest power first)
C = [5.3829749049e-06, 4.8890637117e-05, 3.8003574446e-05, 3.2776263542e-03, 2.1141005680e-02, 4.9867346883e-02]


def gelu_as(x):
    ax = np.abs(x)
    p = np.float64(np.float32(C[0])) * ax + np.float64(np.float32(C[1]))
    for c in C[2:]:
        p = p * ax + np.float64(np.float32(c))
    p = p * ax + 1.0
    return np.maximum(x, 0.0) - 0.5 * ax * p ** -16.0


def test_folded_coefficients_are_abramowitz_stegun_7_1_28():
    a = [0.0705230784, 0.0422820123, 0.0092705272, 0.0001520143, 0.0002765672, 0.0000430638]  # A-S 7.1.28, a1..a6
    folded = [a[k] / np.sqrt(2.0) ** (k + 1) for k in range(6)]
    assert np.allclose(folded[::-1], C, rtol=2e-7, atol=0)


def test_gelu_formula_error_is_far_below_the_bf16_rounding_of_its_output():
    x = np.linspace(-12.0, 12.0, 2_400_001)
    want = 0.5 * x * erfc(-x / np.sqrt(2.0))
    err = np.abs(gelu_as(x) - want)
    assert err.max() < 6e-7, err.max()               # 3e-7 on erfc x 0.5 |x| at |x| ~ 2..4
    # against half an ulp of the bf16 value it is stored as: never more than 1 % of it wherever GELU is not ~ 0
    big = np.abs(want) > 1e-3
    assert (err[big] / (np.abs(want[big]) * 2.0 ** -9)).max() < 0.35
    # saturation: beyond |x| ~ 27 the 16th power overflows fp32, v_rcp_f32(inf) = 0 and the result is max(x, 0) exactly
    p = np.float32(1.0)
    ax = np.float32(30.0)
    with np.errstate(over="ignore"):
        q = np.float32(C[0]) * ax + np.float32(C[1])
        for c in C[2:]:
            q = q * ax + np.float32(c)
        q = q * ax + p
        for _ in range(4):
            q = q * q
    assert np.isinf(q)


def test_relu_as_a_half_sum_is_exact_for_every_normal_fp32():
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 32, size=4_000_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x) & (np.abs(x) >= np.float32(2.0 ** -125))]   # halves of these are representable
    half = np.float32(0.5)
    got = half * x + half * np.abs(x)
    want = np.maximum(x, np.float32(0.0))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for v in (0.0, -0.0, np.inf):
        v = np.float32(v)
        assert (half * v + half * np.abs(v)).view(np.uint32) == np.maximum(v, np.float32(0.0)).view(np.uint32)
    # the one place it is NOT exact: a denormal with an odd last bit loses that bit in the halving (2^-149 -> 0).
    # fc1 pre-activations of that size do not occur, and the difference (<= 2^-149) is far below the bf16 rounding.
    tiny = np.uint32(1).view(np.float32)
    assert half * tiny + half * np.abs(tiny) != tiny


def test_tile_map_multiply_shift_division():
    """csrc/dvt_vit.hip fd_make / fd_div (round 6): the 256 x 256 GEMM's tile map divides by launch-invariant divisors with a
    multiply-high and two shifts (Granlund-Montgomery, round-up form) instead of hipcc's ~30-instruction VALU expansion of
    each run-time division.  The arithmetic, restated: l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1, t = hi32(m x),
    x // d = (t + ((x - t) >> 1)) >> (l - 1) -- exact for every 32-bit x; d = 1 is passed through."""
    import numpy as np

    def fd_make(d):
        l = 0
        while (1 << l) < d:
            l += 1
        return (0 if d <= 1 else (((1 << l) - d) << 32) // d + 1), l

    def fd_div(x, fd):
        m, l = fd
        if l == 0:
            return x
        t = (x.astype(np.uint64) * np.uint64(m)) >> np.uint64(32)
        return ((t + ((x.astype(np.uint64) - t) >> np.uint64(1))) >> np.uint64(l - 1)).astype(np.uint64)

    rng = np.random.RandomState(0)
    xs = np.concatenate([np.arange(0, 70000, dtype=np.uint64), rng.randint(0, 2 ** 31 - 1, 200000).astype(np.uint64),
                         np.array([2 ** 31 - 1, 2 ** 32 - 1], np.uint64)])
    # the divisors the ViT shapes produce (group * mt, mblock * group, group, mblock) and a sweep
    ds = [1, 2, 3, 4, 5, 7, 9, 12, 36, 48, 2139, 2140, 9 * 2139, 12 * 2140, 3 * 2087, 64, 65, 1000003, 2 ** 30 + 1, 2 ** 31 - 1]
    ds += list(rng.randint(1, 2 ** 31 - 1, 50))
    for d in ds:
        fd = fd_make(int(d))
        assert fd[0] < 2 ** 32
        got = fd_div(xs, fd)
        assert np.array_equal(got, xs // np.uint64(d)), d


# ---------------------------------------------------------------------------------------------------------------------------
# 3. The log2-domain attention loop (round 6; csrc/dvt_vit.hip, attention_v2_body with VAR bit 32 = dvt_vit_attention_log2q), stated
#    on the CPU for ONE wave = 16 queries of one (image, head), in the kernel's own fp32 / bf16 arithmetic:
#      * q arrives as bf16(q * log2(e) / 8) -- the qkv GEMM's epilogue rounds ONCE, exactly as it rounds an unscaled q;
#      * a tile's logits leave the matrix pipe as t = k . q' - m (the accumulation starts from C = -m, m = the running max in
#        units of log2), P = 2^t with no further arithmetic;
#      * tile 0: m = the tile's exact max (prologue).  Later tiles: nothing is reduced while every LANE's 16-term partial row
#        sum stays <= 2980 (~ e^8: a lane holds keys 16 mt + 4 g + r of the tile, g = its quarter); otherwise (wave vote) the
#        max grows by d = max(t, 0) per query, o and l are scaled by 2^-d, and t is lowered by d -- here by recomputing the
#        tile, in the kernel also on the already issued next tile;
#      * P is rounded to bf16 for P.V (fp32 accumulation), the row sum l takes the unrounded P; keys >= n_valid are masked.
#    Reference: softmax(q k^T / 8) v (timm Attention.forward under the reference's autocast, vit_wrapper.py:122-143).
def _bf16(x):
    import torch
    return torch.as_tensor(np.asarray(x, dtype=np.float32)).bfloat16().float().numpy()


def _attention_log2_wave(q, k, v, n_valid, stats):
    """q [16, 64], k / v [s, 64] fp32 arrays holding bf16 values -> out [16, 64]; stats counts the redone tiles."""
    qp = _bf16(q * np.float32(0.125 * 1.4426950408889634))
    m = np.zeros(16, np.float32)
    lsum = np.zeros(16, np.float32)
    o = np.zeros((16, 64), np.float32)
    ntiles = (n_valid + 63) // 64
    lane_of_key = (np.arange(64) % 16) // 4  # g of key 16 mt + 4 g + r
    for kt in range(ntiles):
        kk = k[kt * 64:(kt + 1) * 64]
        vv = v[kt * 64:(kt + 1) * 64]
        if kk.shape[0] < 64:  # (the kernel reads the next rows instead; they are masked below)
            kk = np.vstack([kk, np.zeros((64 - kk.shape[0], 64), np.float32)])
            vv = np.vstack([vv, np.zeros((64 - vv.shape[0], 64), np.float32)])
        valid = kt * 64 + np.arange(64) < n_valid
        sp = (qp.astype(np.float64) @ kk.astype(np.float64).T).astype(np.float32)  # fp32 accumulation of exact bf16 products
        t = np.where(valid[None, :], sp - m[:, None], np.float32(-1e30))
        if kt == 0:
            d = t.max(axis=1)
            m = m + d
            t = np.where(valid[None, :], t - d[:, None], np.float32(-1e30))
        with np.errstate(over="ignore"):
            p = np.exp2(t.astype(np.float32)).astype(np.float32)
        part = np.stack([p[:, lane_of_key == g].sum(axis=1) for g in range(4)], axis=1)
        if not np.all(part <= np.float32(2980.0)):  # wave vote; inf / NaN fail the compare
            stats["redone"] += 1
            d = np.maximum(t.max(axis=1), np.float32(0.0))
            alpha = np.exp2(-d).astype(np.float32)
            lsum, o, m = lsum * alpha, o * alpha[:, None], m + d
            t = np.where(valid[None, :], t - d[:, None], np.float32(-1e30))
            p = np.exp2(t.astype(np.float32)).astype(np.float32)
        lsum = lsum + p.sum(axis=1, dtype=np.float32)
        o = o + (_bf16(p).astype(np.float64) @ vv.astype(np.float64)).astype(np.float32)
    return o / lsum[:, None]


def test_attention_log2_domain_loop_statement():
    rng = np.random.default_rng(0)
    redone_late = 0
    for n_valid, spike in [(1370, None), (1370, (21, 2.0)), (1370, (5, 1.5)), (1370, (0, 1.5)), (200, (2, 1.15)), (65, None), (1, None)]:
        s = (n_valid + 63) // 64 * 64
        q = rng.standard_normal((16, 64)).astype(np.float32)
        k = rng.standard_normal((s, 64)).astype(np.float32)
        v = rng.standard_normal((s, 64)).astype(np.float32)
        if spike is not None:  # one key aligned with a few queries: its logit exceeds everything before it by far more than 8
            key = min(64 * spike[0] + 17, n_valid - 1)
            k[key] = 0.0
            for qi in (3, 9):
                k[key] += q[qi] * spike[1]
        q, k, v = _bf16(q), _bf16(k), _bf16(v)
        stats = {"redone": 0}
        got = _attention_log2_wave(q, k, v, n_valid, stats)
        # the reference on the q the kernel is handed: 2^(q' . k) = e^(ln 2 q' . k)
        qp = _bf16(q * np.float32(0.125 * 1.4426950408889634)).astype(np.float64)
        logits = (qp @ k[:n_valid].astype(np.float64).T) * np.log(2.0)
        w = np.exp(logits - logits.max(axis=1, keepdims=True))
        want = (w / w.sum(axis=1, keepdims=True)) @ v[:n_valid].astype(np.float64)
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() < 2e-2 * max(np.abs(want).max(), 1e-30), (n_valid, spike, np.abs(got - want).max())
        if spike is not None and spike[0] > 0 and spike[1] >= 2.0:  # logit ~ 2 |q|^2 / 8 ~ 16 >> the running max + 8
            assert stats["redone"] >= 1, "the spiked tile must take the exact path"
        redone_late += stats["redone"]
        # ... and against softmax(q k^T / 8) v of the UNSCALED bf16 q (what the reference computes): the pre-scaling is one more
        # realisation of q's bf16 rounding, i.e. the same error class as the bf16 rounding of P
        logits0 = (q.astype(np.float64) * 0.125) @ k[:n_valid].astype(np.float64).T
        w0 = np.exp(logits0 - logits0.max(axis=1, keepdims=True))
        want0 = (w0 / w0.sum(axis=1, keepdims=True)) @ v[:n_valid].astype(np.float64)
        assert np.abs(got - want0).max() < 3e-2 * max(np.abs(want0).max(), 1e-30)
    assert redone_late >= 2  # the late-growth branch ran (tile 21 is also the kernel's half tile: 1370 = 21 * 64 + 26)
