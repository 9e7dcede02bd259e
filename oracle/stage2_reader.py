"""Stage-2 reader oracle -- TEST INFRASTRUCTURE ONLY.

Restates how the reference's stage 2 consumes what stage 1 writes
(dvt/dataset/paired_list_dataset.py:27-43; called from main_denoiser.py with
`feat_root = {save_root}/denoised_features/{model}`): for a work-list entry `img_pth`

    denoised = np.load(join(feat_root, img_pth with its extension replaced by ".npy")).squeeze()
    original = np.load(the same path with "denoised_features" -> "raw_features").squeeze()

and entries whose denoised file does not exist are skipped (the reference resamples another index).
"""
from __future__ import annotations

import os

import numpy as np


def paired_paths(feat_root: str, img_pth: str) -> tuple[str, str]:
    ext = os.path.splitext(img_pth)[1]  # :29
    den = os.path.join(feat_root, img_pth.replace(f"{ext}", ".npy"))  # :30
    return den, den.replace("denoised_features", "raw_features")  # :33


def read_pair(feat_root: str, list_line: str):
    """One `__getitem__` (:27-43) without the image; None when the pair is not there yet (:31-32)."""
    img_pth = list_line.strip().split(" ")[0]  # :24
    den_p, raw_p = paired_paths(feat_root, img_pth)
    if not os.path.exists(den_p):
        return None
    return {"original_feats": np.load(raw_p).squeeze(), "denoised_feats": np.load(den_p).squeeze()}
