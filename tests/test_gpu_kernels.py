"""GPU parity of every C-ABI kernel against the CPU oracle on the same seeded inputs.

Tolerances (fp32 path): integer/index work bit-exact; single kernels rel 2e-5 of the
tensor scale (fp32 summation-order differences: f32 MFMA is an fmaf chain, atomics are
unordered); Adam rel 1e-5 per step.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import hashgrid as hg
from oracle.models import NeuralFeatureFieldOracle, SingleImageDenoiserOracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(got, want):
    got = got.detach().double().cpu() if torch.is_tensor(got) else torch.as_tensor(got).double()
    want = want.detach().double().cpu() if torch.is_tensor(want) else torch.as_tensor(want).double()
    return float((got - want).abs().max() / (want.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def L(built_lib):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return built_lib


def _s():
    return torch.cuda.current_stream().cuda_stream


_KEEP = []


def P(t):
    """Device pointer of `t` moved to the GPU; the device copy is kept alive until the end of
    the test module (never take .data_ptr() of a temporary: the caching allocator would hand
    its block to the next temporary before the kernel has run)."""
    t = t.detach().to(DEV).contiguous()
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return t.data_ptr()


def edge_coords(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(n, 2, generator=g)
    lat = torch.linspace(0, 1, 37)
    gy, gx = torch.meshgrid(lat, lat, indexing="ij")
    special = torch.tensor([[0., 0.], [1., 1.], [1., 0.], [0., 1.], [0.5, 0.5], [1., 0.3]])
    return torch.cat([special, torch.stack([gx, gy], -1).reshape(-1, 2), xy]).contiguous()


# ------------------------------------------------------------------------------ hash grid
@pytest.mark.parametrize("cfg", [(16, 8, 16, 1024, 20), (10, 8, 16, 1024, 20), (16, 8, 16, 1024, 12)])
def test_grid_corners_bit_exact(L, cfg):
    from dvt_amd import _lib
    tbl = _lib.grid_table(*cfg)
    otbl = hg.grid_table(*cfg)
    xy = edge_coords(3000)
    n = xy.shape[0]
    idx = torch.empty((n, cfg[0], 4), device=DEV, dtype=torch.int32)
    w = torch.empty((n, cfg[0], 4), device=DEV, dtype=torch.float32)
    assert L.dvt_grid_corners(C.byref(tbl), P(xy), idx.data_ptr(), w.data_ptr(), n, _s()) == 0
    oi, ow = hg.corners(otbl, xy.numpy())
    got_i = idx.cpu().numpy().astype(np.uint32)
    assert np.array_equal(got_i, oi), f"{(got_i != oi).sum()} corner indices differ"
    assert np.array_equal(w.cpu().numpy(), ow), "corner weights must be bit-exact fp32"


def test_grid_fwd_bwd_vs_oracle(L):
    from dvt_amd import _lib
    cfg = (16, 8, 16, 1024, 16)
    tbl, otbl = _lib.grid_table(*cfg), hg.grid_table(*cfg)
    torch.manual_seed(1)
    params = torch.randn(otbl.n_params, requires_grad=True)
    xy = edge_coords(700, seed=3)
    n = xy.shape[0]
    enc_o = hg.encode(otbl, params, xy)
    d_enc = torch.randn(n, 128)
    enc_o.backward(d_enc)
    enc = torch.empty((n, 128), device=DEV)
    assert L.dvt_grid_fwd(C.byref(tbl), P(xy), P(params), enc.data_ptr(), n, _s()) == 0
    assert relerr(enc, enc_o) < 2e-6
    g = torch.zeros(otbl.n_params, device=DEV)
    touched = torch.zeros((otbl.n_entries_total + 31) // 32, device=DEV, dtype=torch.int32)
    assert L.dvt_grid_bwd(C.byref(tbl), P(xy), P(d_enc), g.data_ptr(),
                          touched.data_ptr(), n, _s()) == 0
    assert relerr(g, params.grad) < 2e-5
    # the bitmap marks exactly the entries any sample's corner refers to
    oi, _ = hg.corners(otbl, xy.numpy())
    want = np.zeros(touched.numel() * 32, bool)
    want[oi.reshape(-1)] = True
    bits = np.unpackbits(touched.cpu().numpy().view(np.uint8), bitorder="little").astype(bool)
    assert np.array_equal(bits, want)


# ------------------------------------------------------------------------------ linear
@pytest.mark.parametrize("m,n,k", [(2048, 384, 128), (2048, 768, 384), (2048, 192, 768),
                                   (1369, 768, 384), (100, 80, 36), (64, 4, 4), (333, 132, 80)])
@pytest.mark.parametrize("relu", [0, 1])
def test_linear_fwd_bwd_vs_torch(L, m, n, k, relu):
    torch.manual_seed(m + n + k)
    x, w, b = torch.randn(m, k), torch.randn(n, k) / k ** 0.5, torch.randn(n)
    x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
    y_ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    if relu:
        y_ref = y_ref.relu()
    dy = torch.randn(m, n)
    y_ref.backward(dy.double())
    dx_, dw_, db_ = x.grad, w.grad, b.grad
    dv = lambda t: t.detach().to(DEV).contiguous()
    X, W, B_, DY = dv(x), dv(w), dv(b), dv(dy)
    Y = torch.empty((m, n), device=DEV)
    assert L.dvt_linear_fwd(X.data_ptr(), W.data_ptr(), B_.data_ptr(), Y.data_ptr(), m, n, k, relu, _s()) == 0
    assert relerr(Y, y_ref) < 2e-6, "forward"
    if relu:
        DY = DY * (Y > 0)
    DW, DB, DX = torch.zeros_like(W), torch.zeros_like(B_), torch.empty_like(X)
    assert L.dvt_linear_bwd(DY.data_ptr(), X.data_ptr(), W.data_ptr(), DW.data_ptr(), DB.data_ptr(),
                            DX.data_ptr(), None, m, n, k, _s()) == 0
    assert relerr(DX, dx_) < 2e-6, "dgrad"
    assert relerr(DW, dw_) < 2e-5, "wgrad"
    assert relerr(DB, db_) < 2e-5, "bias grad"


def test_linear_bwd_relu_mask_and_transpose_detection(L):
    """A = I with an ASYMMETRIC weight catches a swapped C/D fragment layout."""
    n = k = 64
    w = torch.arange(n * k, dtype=torch.float32).reshape(n, k) / 100.0
    x = torch.eye(64)
    Y = torch.empty((64, n), device=DEV)
    assert L.dvt_linear_fwd(P(x), P(w), None, Y.data_ptr(), 64, n, k, 0, _s()) == 0
    assert torch.equal(Y.cpu(), w.t().contiguous())
    # dgrad with the fused relu mask
    torch.manual_seed(0)
    dy, W, mask = torch.randn(128, 96), torch.randn(96, 64), torch.randn(128, 64)
    DX = torch.empty((128, 64), device=DEV)
    assert L.dvt_linear_bwd(P(dy), None, P(W), None, None, DX.data_ptr(),
                            P(mask), 128, 96, 64, _s()) == 0
    assert relerr(DX, (dy.double() @ W.double()) * (mask > 0)) < 2e-6


# ------------------------------------------------------------------------------ gathers
def test_gather_scatter_bilinear(L):
    torch.manual_seed(0)
    src = torch.randn(500, 768)
    idx = torch.randint(0, 5000, (2048,), dtype=torch.int32)
    dst = torch.empty((2048, 768), device=DEV)
    assert L.dvt_gather_rows(P(src), P(idx), dst.data_ptr(), 2048, 768, 500, _s()) == 0
    assert torch.equal(dst.cpu(), src[(idx % 500).long()])
    acc = torch.zeros((500, 768), device=DEV)
    upd = torch.randn(2048, 768)
    assert L.dvt_scatter_add_rows(P(upd), P(idx), acc.data_ptr(), 2048, 768, 500, _s()) == 0
    want = torch.zeros(500, 768, dtype=torch.float64).index_add_(0, (idx % 500).long(), upd.double())
    assert relerr(acc, want) < 1e-5
    # bilinear == F.grid_sample(align_corners=True), generic coords + exact lattice coords
    H, W, Cc = 7, 9, 64
    G = torch.randn(1, Cc, H, W, requires_grad=True)
    lat = torch.stack(torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")[::-1], -1)
    coords = torch.cat([torch.rand(300, 2) * 2 - 1, lat.reshape(-1, 2), torch.tensor([[-1., -1.], [1., 1.]])])
    n = coords.shape[0]
    ref = torch.nn.functional.grid_sample(G, coords[None, None], mode="bilinear", align_corners=True)
    ref = ref.squeeze().permute(1, 0)
    dout = torch.randn(n, Cc)
    ref.backward(dout)
    rows = G.detach().permute(0, 2, 3, 1).reshape(H * W, Cc).contiguous().to(DEV)
    out = torch.empty((n, Cc), device=DEV)
    assert L.dvt_bilinear_rows_fwd(rows.data_ptr(), P(coords), out.data_ptr(), n, Cc, H, W, _s()) == 0
    assert relerr(out, ref) < 2e-6
    dG = torch.zeros((H * W, Cc), device=DEV)
    assert L.dvt_bilinear_rows_bwd(P(dout), P(coords), dG.data_ptr(), n, Cc, H, W, _s()) == 0
    assert relerr(dG, G.grad.permute(0, 2, 3, 1).reshape(H * W, Cc)) < 1e-5


# ------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("c", [768, 1024, 64])
@pytest.mark.parametrize("with_res", [False, True])
def test_loss_fwd_bwd_vs_oracle(L, c, with_res):
    torch.manual_seed(c + with_res)
    n, lattice = 512, 37
    Fm = torch.randn(n, c, requires_grad=True)
    G = (torch.randn(lattice, c) * 0.5).requires_grad_(True)
    raw = torch.randn(n, c) * 2
    Hm = (torch.randn(n, c) * 0.3).requires_grad_(True) if with_res else None
    gi = torch.randint(0, 10 * lattice, (n,), dtype=torch.int32)
    g = G[(gi % lattice).long()]
    pred = Fm + g + (Hm.detach() if with_res else 0)
    l2 = torch.nn.functional.mse_loss(pred, raw)
    cos = 1 - torch.nn.functional.cosine_similarity(pred, raw, dim=-1).mean()
    loss = l2 + cos
    rl = sp = torch.zeros(())
    if with_res:
        rl = 0.1 * torch.nn.functional.mse_loss(Hm, (raw - Fm - g).detach())
        sp = 0.02 * Hm.abs().mean()
        loss = loss + rl + sp
    (loss * 1024.0).backward()
    dF, dH = torch.empty((n, c), device=DEV), torch.empty((n, c), device=DEV)
    rows, out = torch.empty((n, 8), device=DEV), torch.zeros(8, device=DEV)
    assert L.dvt_loss_fwd_bwd(P(Fm), P(G), P(gi), lattice,
                              P(Hm) if with_res else None, P(raw), dF.data_ptr(),
                              dH.data_ptr() if with_res else None, rows.data_ptr(), n, c, 1024.0, _s()) == 0
    assert L.dvt_loss_reduce(rows.data_ptr(), out.data_ptr(), n, c, int(with_res), _s()) == 0
    want = torch.stack([loss, l2, cos, rl, sp]).detach()
    assert relerr(out[:5], want) < 5e-6, (out[:5].cpu(), want)
    assert relerr(dF, Fm.grad) < 1e-5
    if with_res:
        assert relerr(dH, Hm.grad) < 1e-5
    # the G gradient is the row scatter of d_pred
    dG = torch.zeros((lattice, c), device=DEV)
    assert L.dvt_scatter_add_rows(dF.data_ptr(), P(gi), dG.data_ptr(), n, c, lattice, _s()) == 0
    assert relerr(dG, G.grad) < 1e-5


# ------------------------------------------------------------------------------ Adam
def test_adam_dense_sparse_vs_torch(L):
    from dvt_amd import _lib
    torch.manual_seed(0)
    n_sparse, n_dense = 256 * 40, 256 * 12
    n = n_sparse + n_dense
    p0 = torch.randn(n) * 1e-2
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=0.01, eps=1e-15, weight_decay=1e-5, betas=(0.9, 0.99))
    P, M, V, G = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    touched = torch.zeros(n_sparse // 256, device=DEV, dtype=torch.int32)
    lrs = [0.0, 0.0025, 0.005, 0.01, 0.0073, 0.004, 0.002]
    for t, lr in enumerate(lrs, start=1):
        g = torch.zeros(n)
        ent = torch.randperm(n_sparse // 8)[:97]  # sparse entries of 8 floats
        for e in ent.tolist():
            g[e * 8:(e + 1) * 8] = torch.randn(8) * 1024 * 1e-3
        g[n_sparse:] = torch.randn(n_dense) * 1024 * 1e-3
        for grp in opt.param_groups:
            grp["lr"] = lr
        p.grad = g.clone()
        opt.step()
        G.copy_(g.to(DEV))
        bits = np.zeros(n_sparse // 8, np.uint8)
        bits[ent.numpy()] = 1
        touched.copy_(torch.from_numpy(np.packbits(bits, bitorder="little").view(np.int32)).to(DEV))
        a = _lib.AdamArgs()
        a.beta1, a.beta2, a.eps, a.weight_decay = 0.9, 0.99, 1e-15, 1e-5
        a.n_segs, a.sparse_end = 1, n_sparse
        a.segs[0].begin, a.segs[0].end, a.segs[0].lr, a.segs[0].active = 0, n, lr, 1
        a.segs[0].bias_correction1 = 1 - 0.9 ** t
        a.segs[0].bias_correction2_sqrt = (1 - 0.99 ** t) ** 0.5
        assert L.dvt_adam_step(C.byref(a), P.data_ptr(), M.data_ptr(), V.data_ptr(), G.data_ptr(),
                               touched.data_ptr(), _s()) == 0
        st = opt.state[p]
        assert relerr(P, p) < 1e-5, f"step {t} params"
        assert relerr(M, st["exp_avg"]) < 1e-5 and relerr(V, st["exp_avg_sq"]) < 1e-5
        assert float(G.abs().max()) == 0.0 and int(touched.abs().max()) == 0, "zero_grad invariant"
    # untouched entries still moved (dense semantics, quirk Q2)
    assert float((P.cpu() - p0).abs().min()) > 0


# ------------------------------------------------------------------------------ module API
def test_module_api_matches_oracle(L):
    """Reference-style usage: SingleImageDenoiser(...)(raw, coords, NeuralFeatureField, sac);
    loss.backward() -- same numbers as the oracle restatement with identical parameters."""
    from dvt_amd.models import NeuralFeatureField, SingleImageDenoiser
    torch.manual_seed(0)
    C_, H, W, n = 64, 9, 9, 512
    kw = dict(feat_dim=C_, n_levels=8, max_resolution=256, log2_hashmap_size=12)
    f_o = NeuralFeatureFieldOracle(**kw)
    d_o = SingleImageDenoiserOracle(H, W, C_, 3)
    with torch.no_grad():
        f_o.neural_field.params.normal_(0, 0.1)
    f_h = NeuralFeatureField(**kw)
    d_h = SingleImageDenoiser(H, W, C_, 3)
    f_h.load_state_dict(f_o.state_dict())
    d_h.load_state_dict(d_o.state_dict())
    f_h, d_h = f_h.to(DEV), d_h.to(DEV)
    raw, xy = torch.randn(n, C_), torch.rand(n, 2)
    lat = torch.stack(torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")[::-1], -1).reshape(-1, 2)
    sac = lat[torch.randint(0, H * W, (n,))]
    for phase2 in (False, True):
        if phase2:
            for d in (d_o, d_h):
                d.stop_shared_artifacts_grad(); d.start_residual_predictor()
        for mods in ((d_o, f_o), (d_h, f_h)):
            for mm in mods:
                mm.zero_grad(set_to_none=True)
        out_o = d_o(raw, xy, f_o, sac)
        (out_o["loss"] * 1024).backward()
        out_h = d_h(raw.to(DEV), xy.to(DEV), f_h, sac.to(DEV))
        (out_h["loss"] * 1024).backward()
        assert set(out_h) == set(out_o)
        for k in out_o:
            assert relerr(out_h[k], out_o[k]) < 1e-5, k
        assert relerr(f_h.neural_field.params.grad, f_o.neural_field.params.grad) < 2e-5
        for (ka, pa), (kb, pb) in zip(list(f_h.mlp.named_parameters()) + list(d_h.named_parameters()),
                                      list(f_o.mlp.named_parameters()) + list(d_o.named_parameters())):
            assert ka == kb
            if pb.grad is None:
                assert pa.grad is None or float(pa.grad.abs().max()) == 0, ka
            else:
                assert relerr(pa.grad, pb.grad) < 3e-5, ka
    with torch.no_grad():
        rawv, xyv = torch.randn(1, H, W, C_), torch.rand(1, H, W, 2)
        vo = d_o(rawv, xyv, f_o, return_visualization=True)
        vh = d_h(rawv.to(DEV), xyv.to(DEV), f_h, return_visualization=True)
    assert set(vo) == set(vh)
    for k in vo:
        assert relerr(vh[k], vo[k]) < 1e-5, k
