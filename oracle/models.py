"""Oracle restatements of the reference's stage-1 models (plain PyTorch, CPU, fp32/fp64).

  NeuralFeatureFieldOracle  dvt/models/neural_feature_field.py:11-49
  SingleImageDenoiserOracle dvt/models/offline_denoiser.py:11-171 (pinned against the
                            imported reference class by tests/golden/denoiser_*.npz)
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .hashgrid import HashGridOracle


class NeuralFeatureFieldOracle(nn.Module):
    def __init__(self, feat_dim=768, base_resolution=16, max_resolution=1024, n_levels=10,
                 n_features_per_level=8, log2_hashmap_size=20):
        super().__init__()
        # neural_feature_field.py:25-39
        self.neural_field = HashGridOracle(n_levels, n_features_per_level, base_resolution,
                                           max_resolution, log2_hashmap_size)
        # :40-44
        self.mlp = nn.Sequential(
            nn.Linear(self.neural_field.n_output_dims, feat_dim // 2),
            nn.ReLU(),
            nn.Linear(feat_dim // 2, feat_dim),
        )

    def forward(self, coords):
        # :46-49
        assert coords.max() <= 1 and coords.min() >= 0, "coordinates should be in [0, 1]"
        enc = self.neural_field(coords.reshape(-1, 2))
        return self.mlp(enc.view(list(coords.shape[:-1]) + [-1]))


class SingleImageDenoiserOracle(nn.Module):
    def __init__(self, noise_map_height=37, noise_map_width=37, feat_dim=768, layer_index=11,
                 enable_residual_predictor=True, disable_pe=False):
        super().__init__()
        self.feat_dim = feat_dim
        # offline_denoiser.py:27-36
        if disable_pe:
            self.shared_artifacts = nn.Parameter(
                torch.zeros(1, feat_dim, noise_map_height, noise_map_width), requires_grad=False)
        else:
            self.shared_artifacts = nn.Parameter(
                torch.randn(1, feat_dim, noise_map_height, noise_map_width) * 0.02)
        self.enable_residual_predictor = enable_residual_predictor
        if enable_residual_predictor:  # :38-46
            self.residual_predictor = nn.Sequential(
                nn.Linear(feat_dim, feat_dim // 4), nn.ReLU(),
                nn.Linear(feat_dim // 4, feat_dim // 4), nn.ReLU(),
                nn.Linear(feat_dim // 4, feat_dim))
        self.residual_predictor_start = False

    def start_residual_predictor(self):  # :49-51
        self.residual_predictor_start = True

    @property
    def use_residual_predictor(self):  # :53-56
        return self.enable_residual_predictor and self.residual_predictor_start

    def stop_shared_artifacts_grad(self):  # :58-60
        self.shared_artifacts.requires_grad = False

    def forward(self, raw_vit_outputs, global_pixel_coords, neural_field=None,
                shared_artifact_coords=None, return_visualization=False) -> Dict[str, torch.Tensor]:
        if raw_vit_outputs.dim() != 2:  # :86-92
            original_shape = raw_vit_outputs.shape
            raw_vit_outputs = raw_vit_outputs.reshape(-1, self.feat_dim)
            global_pixel_coords = global_pixel_coords.reshape(-1, 2)
            shared = self.shared_artifacts.permute(0, 2, 3, 1).reshape(-1, self.feat_dim)
        else:  # :93-102
            original_shape = None
            shared = F.grid_sample(self.shared_artifacts, shared_artifact_coords[None, None, ...],
                                   mode="bilinear", align_corners=True)
            shared = shared.squeeze().permute(1, 0)
        feats = neural_field(global_pixel_coords)  # :104
        res = self.residual_predictor(raw_vit_outputs) if self.use_residual_predictor else None
        pred = feats + shared + res.detach() if res is not None else shared + feats  # :113-118
        l2 = F.mse_loss(pred, raw_vit_outputs)  # :122
        cos = 1 - F.cosine_similarity(pred, raw_vit_outputs, dim=-1).mean()  # :123-124
        loss = l2 + cos
        out = {"patch_l2_loss": l2, "loss": loss, "cosine_similarity_loss": cos}
        if res is not None:  # :131-140
            gt = (raw_vit_outputs - feats - shared).detach()
            rl = 0.1 * F.mse_loss(res, gt)
            sp = 0.02 * res.abs().mean()
            loss = loss + rl + sp
            out.update(loss=loss, residual_loss=rl, residual_sparsity_loss=sp)
        if return_visualization:  # :142-169
            shp = tuple(original_shape[:-1]) + (-1,)
            out["raw_vit_outputs"] = raw_vit_outputs.detach().reshape(shp)
            out["pred_features"] = pred.detach().reshape(shp)
            out["denoised_feats"] = feats.detach().reshape(shp)
            out["shared_patterns"] = shared.detach().reshape(shp)
            if res is not None:
                out["pred_residual"] = res.detach().reshape(shp)
                out["shared_patterns_and_residual"] = (shared + res).detach().reshape(shp)
                out["denoised_features"] = (raw_vit_outputs - shared - res).detach().reshape(shp)
            else:
                out["denoised_features"] = (raw_vit_outputs - shared).detach().reshape(shp)
        return out
