"""Feature-provider side of the call ``naf(image, lr_features, target_size)``.

The reference obtains ``lr_features`` from a vision foundation model through ``PretrainedViTWrapper``
(src/backbone/vit_wrapper.py:46-180: timm / torch.hub backbones, ``forward`` returns the last block's patch tokens as an
NCHW map, prefix tokens dropped; patch size parsed from the model name, :70-79).  Backbones are out of scope here
(no timm, no network), but the hand-over format is part of the hot path's boundary, so this
module provides
  * ``tokens_to_feature_map``: ViT token sequence [B, prefix + h*w, C] -> [B, C, h, w] exactly as the wrapper's
    ``rearrange(out, "b (h w) c -> b c h w")`` (vit_wrapper.py:162) does, as a zero-copy channels-last view (the layout
    ``naf_pack_values`` reads fastest);
  * ``FeatureProvider``: the protocol a backbone wrapper has to satisfy to be used with ``NAF`` (``patch_size``,
    ``embed_dim``, ``forward(image) -> [B, C, H/ps, W/ps]``);
  * ``SyntheticViT``: a random-weight stand-in with the interface and the output shape / layout of a ViT-S/B/L
    (patch embedding + prefix tokens), for tests and benchmarks without a network -- BASELINE.json's configurations use
    random features of exactly these shapes.
"""
from __future__ import annotations

from typing import Protocol, Tuple, runtime_checkable

import torch
from torch import nn


def tokens_to_feature_map(tokens: torch.Tensor, grid: Tuple[int, int], num_prefix_tokens: int = 0) -> torch.Tensor:
    """[B, P + h*w, C] tokens (class / register tokens first, as timm orders them) -> logical [B, C, h, w].

    The result is a view of the token buffer: its memory is channels-last, which ``naf_pack_values`` copies with full
    16-byte accesses.  Mirrors vit_wrapper.py:159-162."""
    h, w = int(grid[0]), int(grid[1])
    if tokens.dim() != 3 or tokens.shape[1] != num_prefix_tokens + h * w:
        raise ValueError(f"expected [B, {num_prefix_tokens} + {h}*{w}, C] tokens, got {tuple(tokens.shape)}")
    patch = tokens[:, num_prefix_tokens:]
    return patch.unflatten(1, (h, w)).permute(0, 3, 1, 2)


@runtime_checkable
class FeatureProvider(Protocol):
    """What ``NAF`` needs from a backbone wrapper (the attributes eval_seg_probing.py:94-135 reads from the reference's)."""
    patch_size: int
    embed_dim: int

    def __call__(self, image: torch.Tensor) -> torch.Tensor: ...


class SyntheticViT(nn.Module):
    """Random-weight patch embedder with a ViT's token interface: image [B, 3, H, W] -> tokens [B, P + h*w, C] ->
    features [B, C, h, w] (h = H // patch_size).  No attention blocks: it exists to exercise the hand-over format."""

    def __init__(self, embed_dim: int = 384, patch_size: int = 14, num_prefix_tokens: int = 1, seed: int = 0):
        super().__init__()
        self.patch_size, self.embed_dim, self.num_prefix_tokens = patch_size, embed_dim, num_prefix_tokens
        g = torch.Generator().manual_seed(seed)
        self.proj = nn.Conv2d(3, embed_dim, patch_size, stride=patch_size)
        with torch.no_grad():
            self.proj.weight.copy_(torch.randn(self.proj.weight.shape, generator=g) / (3 * patch_size * patch_size) ** 0.5)
            self.proj.bias.zero_()
        self.prefix = nn.Parameter(torch.randn(1, num_prefix_tokens, embed_dim, generator=g), requires_grad=False)

    @torch.no_grad()
    def forward_tokens(self, image: torch.Tensor) -> Tuple[torch.Tensor, Tuple[int, int]]:
        x = self.proj(image.float())                                   # [B, C, h, w]
        B, C, h, w = x.shape
        tok = x.flatten(2).transpose(1, 2)                              # b (h w) c
        tok = torch.cat([self.prefix.expand(B, -1, -1).to(tok.dtype), tok], dim=1)
        return tok, (h, w)

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        tok, grid = self.forward_tokens(image)
        return tokens_to_feature_map(tok, grid, self.num_prefix_tokens)
