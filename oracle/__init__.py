"""CPU oracle for the DVT stage-1 hot path -- TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this package, and only as the checker; the product (`denoising-vit_amd/`) never does.

Each function restates one piece of the reference (Jiawei-Yang/Denoising-ViT) in plain
PyTorch / numpy and cites the file:line it follows.

PARITY PINNING
  * pinned against the reference itself (imported in the build container, fixtures in
    tests/golden/, generator tests/golden/make_golden.py): `SingleImageDenoiser.forward`
    (offline_denoiser.py), `adjust_learning_rate` (misc.py), the index stream, Adam
    (torch.optim.Adam IS the reference's optimizer).
  * PARITY UNPINNED: the hash-grid encoding (third-party tiny-cuda-nn, git master, not in
    the reference tree and CUDA-only) and the ViT forward (timm 1.0.7, not installed, no
    weights).  Their restatements follow the published algorithms (see hashgrid.py, vit.py);
    the ViT is cross-checked against the independent `transformers` Dinov2Model.
"""
