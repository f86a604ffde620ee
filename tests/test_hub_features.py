"""CPU tests of the torch.hub entry's checkpoint path and of the feature hand-over format."""
import pytest
import torch

from oracle import naf_oracle as O


def test_hub_loads_a_reference_keyed_checkpoint_from_a_file_url(tmp_path, monkeypatch):
    """hubconf.naf(pretrained=True) goes through torch.hub.load_state_dict_from_url like the reference (hubconf.py:21-23);
    a file:// URL exercises that path without a network.  Keys are the reference's (strict load)."""
    import hubconf
    p = O.make_params(seed=33)
    ckpt = tmp_path / "naf_release.pth"
    torch.save({k: v.clone() for k, v in p.items()}, ckpt)
    monkeypatch.setattr(hubconf, "CHECKPOINT_URL", ckpt.as_uri())
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "hub"))
    m = hubconf.naf(pretrained=True, device="cpu")
    sd = m.state_dict()
    assert set(sd) == set(p) and all(torch.equal(sd[k], p[k]) for k in p)
    assert sorted(k for k in sd if "rope" in k) == ["image_encoder.rope.periods"]
    torch.save({k: v for k, v in p.items() if "sem_encoder.0" not in k}, tmp_path / "bad.pth")
    monkeypatch.setattr(hubconf, "CHECKPOINT_URL", (tmp_path / "bad.pth").as_uri())
    with pytest.raises(RuntimeError, match="Missing key"):
        hubconf.naf(pretrained=True, device="cpu")


def test_tokens_to_feature_map_is_the_wrappers_rearrange():
    """vit_wrapper.py:159-162: "b (h w) c -> b c h w" after dropping the prefix tokens; here a zero-copy channels-last view."""
    from naf_amd import features
    B, C, h, w, P = 2, 24, 5, 7, 5
    tok = torch.arange(B * (P + h * w) * C, dtype=torch.float32).view(B, P + h * w, C)
    f = features.tokens_to_feature_map(tok, (h, w), P)
    assert f.shape == (B, C, h, w) and f.data_ptr() == tok[:, P:].data_ptr()
    import einops
    assert torch.equal(f, einops.rearrange(tok[:, P:], "b (h w) c -> b c h w", h=h, w=w))
    with pytest.raises(ValueError):
        features.tokens_to_feature_map(tok, (h, w), P + 1)


def test_synthetic_vit_satisfies_the_provider_protocol():
    from naf_amd import features
    v = features.SyntheticViT(embed_dim=64, patch_size=14, num_prefix_tokens=5, seed=1)
    assert isinstance(v, features.FeatureProvider) and v.patch_size == 14 and v.embed_dim == 64
    x = torch.randn(2, 3, 56, 70)
    f = v(x)
    assert f.shape == (2, 64, 4, 5)
    tok, grid = v.forward_tokens(x)
    assert tok.shape == (2, 5 + 20, 64) and grid == (4, 5)
    assert torch.equal(v(x), f)                                   # deterministic
