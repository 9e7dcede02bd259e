"""The developer build of the library (csrc/lab/, -DDVT_LAB -> csrc/libdvt_hip_lab.so, `python tools/build_lab.py`) held to the
same checks as the product kernels: superseded GEMM schedules (0, 2), the 8p re-schedules (5 "8m", 10 "8h": bit-identical),
the 4-wave persistent GEMM (6..9), the round-2 attention loop and the other attention schedule masks.  The module's
fixture (re)builds that library when it is missing or stale (`__graft_entry__.build()` builds it too, after the product
library; nothing under denoising-vit_amd/dvt_amd loads it); without it and without a working hipcc these tests are skipped."""
import pytest
import torch
import torch.nn.functional as F

from tests import test_gpu_vit as V
from tests.test_gpu_vit import DEV, GEMM_DEFAULT, _s

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import dvt_amd.vit  # noqa: F401 registers signatures
    from dvt_amd import _lib
    try:  # (re)built when missing or stale: about a minute, the 4-wave kernel's ablation builds dominate
        _lib.build(lab=True)
    except Exception as exc:  # no hipcc on this box: the product suite does not depend on the developer library
        pytest.skip(f"developer library could not be built ({exc!r}); python tools/build_lab.py")
    h = _lib.open_library(_lib.LAB_LIB_PATH)
    assert h.dvt_vit_is_lab_build() == 1
    return h


@pytest.mark.parametrize("variant", [0, 2, 5, 10, 13])
@pytest.mark.parametrize("m,n,k", V.RESID_SHAPES)
def test_lab_gemm_residual_vs_torch(L, m, n, k, variant):
    V.check_gemm_residual_vs_torch(L, m, n, k, variant)


@pytest.mark.parametrize("variant", [0, 2, 5, 6, 7, 10, 13])
def test_lab_gemm_bias_variants(L, variant):
    V.check_gemm_bias_variants(L, variant)


@pytest.mark.parametrize("attn_variant", V.LAB_ATTN_CASES)
@pytest.mark.parametrize("batch,heads,s_pad,n_valid", V.ATTN_SHAPES)
def test_lab_attention_vs_torch(L, batch, heads, s_pad, n_valid, attn_variant):
    V.check_attention_vs_torch(L, batch, heads, s_pad, n_valid, attn_variant)


@pytest.mark.parametrize("attn_variant", V.LAB_ATTN_CASES)
@pytest.mark.parametrize("spike_tile,gain", V.SPIKES)
def test_lab_attention_late_max_growth(L, attn_variant, spike_tile, gain):
    V.check_attention_late_max_growth(L, attn_variant, spike_tile, gain)


def test_lab_ablation_state_is_reset_by_a_schedule_change(L):
    """the ablation mask of the 4w kernel and the timing build of the 8p kernel were ONE sticky global; after stamps
    (5, -303) a switch to schedule 6 / 7 ran a wrong-result ablation silently.  Now they are separate, bound to the schedule
    selected when they are set, and every change of schedule clears them."""
    try:
        assert L.dvt_tune_set(1, 4) == 0 and L.dvt_tune_set(1, -303) == -1   # schedule 4 has no timing build
        assert L.dvt_tune_set(1, 5) == 0 and L.dvt_tune_set(1, -303) == 0
        assert L.dvt_tune_set(1, 7) == 0                                     # clears the 8p timing build ...
        V.check_gemm_bias_variants(L, 7)                                     # ... so 4w computes RIGHT results
        assert L.dvt_tune_set(1, 7) == 0 and L.dvt_tune_set(1, -303) == 0    # a 4w ablation (wrong results, timing only) ...
        assert L.dvt_tune_set(1, 4) == 0                                     # ... does not survive the switch back
        V.check_gemm_bias_variants(L, 4)
        for v in (-364, -399, -450, -499):                                    # retired 8q values
            assert L.dvt_tune_set(1, v) == -1, v
    finally:
        L.dvt_tune_set(1, GEMM_DEFAULT)
        L.dvt_tune_set(1, -300)


@pytest.mark.parametrize("variant", [5, 10, 11, 13])
@pytest.mark.parametrize("n,k,gelu", [(2304, 768, 0), (3072, 768, 1), (768, 3072, 0)])
def test_gemm_8m_8h_bit_identical_to_8p(L, n, k, gelu, variant):
    """The walks of the 8p ring.  Schedule 4 -- the product's -- is since round 6 the "8b" walk (fragment reads balanced
    8 / 4 / 8 / 4 over the phases, B0 of the next k-tile prefetched in P4, literal ring parity, buffer-descriptor DMA);
    dvt_tune_set(1, 13) is round 5's walk (12 / 4 / 8 / 0 reads, run-time parity), and two re-schedules of THAT one: dvt_tune_set(1, 5) stages every half-tile in the middle of its phase's MFMA segment
    (after the phase's counted wait instead of before it; waits one stage tighter); dvt_tune_set(1, 10) walks a k-tile in two
    phases of 32 MFMAs instead of four of 16 (half the barriers, its own counted waits).  Same MFMAs in the same k order on
    the same operands, so the output must equal the 8p kernel's BIT FOR BIT -- at a size that keeps every CU busy for many
    rounds of tiles (a slot re-staged or read too early shows up as a different bit somewhere), five launches in a row."""
    m = 256 * 520
    g = torch.Generator(device=DEV).manual_seed(n + k)
    x = (torch.rand(m, k, device=DEV, generator=g) * 2 - 1).bfloat16()
    w = ((torch.rand(n, k, device=DEV, generator=g) * 2 - 1) / k ** 0.5 * 1.7).bfloat16()
    b = torch.randn(n, device=DEV, generator=g)
    outs = {}
    try:
        for v, reps in ((4, 1), (variant, 5)):
            assert L.dvt_tune_set(1, v) == 0
            for r in range(reps):
                y = torch.full((m, n), float("nan"), device=DEV, dtype=torch.bfloat16)
                assert L.dvt_vit_gemm_lnfold(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), m, n, k, None, None, gelu, _s()) == 0
                torch.cuda.synchronize()
                outs[(v, r)] = y
    finally:
        L.dvt_tune_set(1, GEMM_DEFAULT)
    ref = outs[(4, 0)].view(torch.int16)
    assert bool(torch.isfinite(outs[(4, 0)].float()).all())
    for r in range(5):
        assert torch.equal(outs[(variant, r)].view(torch.int16), ref), (variant, n, k, r)


@pytest.mark.parametrize("variant,grid", [(4, 0), (5, 0), (10, 0), (6, 0), (7, 0), (7, 1), (7, 3), (7, 5), (9, 2)])
@pytest.mark.parametrize("m,n,k,gelu,fold", [(2048, 1024, 768, 1, 1), (1280, 3072, 768, 1, 1), (1536, 2304, 768, 0, 0),
                                             (1024, 512, 1024, 0, 0), (768, 768, 3072, 0, 0)])
def test_gemm_4w_persistent_vs_fp64(L, m, n, k, gelu, fold, variant, grid):
    """The 4-wave persistent GEMM with the deferred epilogue (dvt_tune_set(1, 6 .. 9), csrc/dvt_vit_gemm4w.inc) through the
    fc1-type entry point dvt_vit_gemm_lnfold -- folded LayerNorm + GELU, or the bias epilogue -- against fp64, next to the
    default 8p kernel (variant 4) on the same operands.  `grid` forces the number of workgroups (dvt_tune_set(1, -600 - n)),
    so that a workgroup runs SEVERAL tiles: the parked tile drains under the next tile's k-loop, the ring runs through the
    tile boundary, the last tile is flushed after the loop.  Every element is compared; a second launch must reproduce the
    first bit for bit (the kernel has no atomics and no data-dependent order).  Variant 9 uses the opt-in cheaper GELU
    (2.7e-4 max abs deviation from erf-GELU before the bf16 rounding): looser bound."""
    g = torch.Generator(device=DEV).manual_seed(m + n + k)
    x = (torch.rand(m, k, device=DEV, generator=g) * 2 - 1 +
         torch.linspace(-1, 1, k, device=DEV)[None, :] * torch.linspace(0.5, 2, m, device=DEV)[:, None]).bfloat16()
    w = ((torch.rand(n, k, device=DEV, generator=g) * 2 - 1) / k ** 0.5 * 1.7 +
         torch.linspace(-0.02, 0.03, n, device=DEV)[:, None]).bfloat16()
    b = torch.randn(n, device=DEV, generator=g)
    stats = cs = None
    acc = x.double() @ w.double().t()
    if fold:
        stats = torch.stack([torch.randn(m, device=DEV, generator=g) * 0.3, torch.rand(m, device=DEV, generator=g) + 0.5], 1).contiguous()
        cs = w.float().sum(1).contiguous()
        acc = stats[:, 1:2].double() * (acc - stats[:, 0:1].double() * cs.double()[None, :])
    want = acc + b.double()
    if gelu:
        want = F.gelu(want.float().bfloat16().double())  # the reference's autocast semantics: GELU of the bf16 linear output
    outs = []
    try:
        assert L.dvt_tune_set(1, variant) == 0 and L.dvt_tune_set(1, -600 - grid) == 0
        for _ in range(2):
            y = torch.full((m, n), float("nan"), device=DEV, dtype=torch.bfloat16)
            assert L.dvt_vit_gemm_lnfold(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), m, n, k,
                                         stats.data_ptr() if fold else None, cs.data_ptr() if fold else None, gelu, _s()) == 0
            torch.cuda.synchronize()
            outs.append(y)
    finally:
        L.dvt_tune_set(1, GEMM_DEFAULT)
        L.dvt_tune_set(1, -600)
    y = outs[0]
    assert bool(torch.isfinite(y.float()).all())
    err = float((y.double() - want).abs().max() / want.abs().max())
    assert err < (8e-3 if variant >= 8 else 6e-3), (variant, grid, (m, n, k), err)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
