"""backbones/eva/rotar_embedding_cat.py of the reference: the rotary embedding table RotaryEmbeddingCat produces for a (H, W) token grid
(freq_bands :35-47, build_fourier_pos_embed :50-112, build_rotary_pos_embed :137-171, RotaryEmbeddingCat :174-253).  The table is constant data of
the geometry: it is built on the host in float32 with the reference's operation order and cached per grid on the device; applying it (rot /
apply_rot_embed_cat, :117-135) is the C ABI's iseg_qkv_rope kernel (functional.qkv_rope)."""
import math

import torch

from ... import nn


def pixel_freq_bands(num_bands, max_freq=224.0, linear_bands=True):
    if linear_bands:
        bands = torch.linspace(1.0, max_freq / 2, num_bands, dtype=torch.float32)
    else:
        bands = torch.pow(torch.tensor(2.0), torch.linspace(0.0, math.log(max_freq) / math.log(2) - 1, num_bands, dtype=torch.float32))
    return bands * math.pi


def freq_bands(num_bands, temperature=10000.0, step=2):
    rg = torch.arange(0, num_bands, step, dtype=torch.float32)
    return 1.0 / (temperature ** (rg / num_bands))


def build_fourier_pos_embed(feat_shape, bands=None, num_bands=64, max_res=224, temperature=10000.0, linear_bands=False, in_pixels=True,
                            ref_feat_shape=None):
    if bands is None:
        bands = pixel_freq_bands(num_bands, float(max_res), linear_bands) if in_pixels else freq_bands(num_bands, temperature, step=1)
    if in_pixels:
        t = [torch.linspace(-1.0, 1.0, s, dtype=torch.float32) for s in feat_shape]
    else:
        t = [torch.arange(s, dtype=torch.float32) for s in feat_shape]
    if ref_feat_shape is not None:
        t = [x / f * r for x, f, r in zip(t, feat_shape, ref_feat_shape)]
    grid = torch.stack(torch.meshgrid(*t, indexing="ij"), dim=-1).unsqueeze(-1)      # [H, W, 2, 1]
    pos = grid * bands                                                                # [H, W, 2, num_bands]
    return torch.sin(pos), torch.cos(pos)


def build_rotary_pos_embed(feat_shape, bands=None, filters=64, max_res=224, temperature=10000, linear_bands=False, in_pixels=True,
                           ref_feat_shape=None):
    sin_emb, cos_emb = build_fourier_pos_embed(feat_shape, bands=bands, num_bands=filters // 4, max_res=max_res, temperature=temperature,
                                               linear_bands=linear_bands, in_pixels=in_pixels, ref_feat_shape=ref_feat_shape)
    n = 1
    for s in feat_shape:
        n *= int(s)
    sin_emb = sin_emb.reshape(n, -1).repeat_interleave(2, dim=-1)      # tf.repeat(repeats=[2], axis=-1): every band twice, side by side
    cos_emb = cos_emb.reshape(n, -1).repeat_interleave(2, dim=-1)
    return sin_emb, cos_emb


class RotaryEmbeddingCat:
    """call(spatial_size) -> fp32 [H W, 2 * filters] = [sin | cos] on the compute device"""

    def __init__(self, filters, max_res=224, temperature=10000.0, in_pixels=True, linear_bands=False, feat_shape=None, ref_feat_shape=None):
        self.filters, self.max_res, self.temperature = int(filters), max_res, temperature
        self.in_pixels, self.linear_bands = in_pixels, linear_bands
        self.feat_shape, self.ref_feat_shape = feat_shape, ref_feat_shape
        if feat_shape is None:
            self.bands = (pixel_freq_bands(self.filters // 4, float(max_res), linear_bands) if in_pixels
                          else freq_bands(self.filters // 4, temperature, step=1))
            self.pos_embed = None
        else:
            self.bands = None
            self.pos_embed = torch.cat(build_rotary_pos_embed(feat_shape, filters=self.filters, max_res=max_res, linear_bands=linear_bands,
                                                              in_pixels=in_pixels, ref_feat_shape=ref_feat_shape), dim=-1)
        self._cache = {}

    def get_embed_host(self, shape=None):
        if self.bands is not None and shape is not None:
            return torch.cat(build_rotary_pos_embed(list(shape), self.bands, in_pixels=self.in_pixels, ref_feat_shape=self.ref_feat_shape), dim=-1)
        if self.pos_embed is not None:
            return self.pos_embed
        raise ValueError("get_embed() requires pre-computed pos_embed or valid shape w/ pre-computed bands")

    def __call__(self, spatial_size):
        key = (tuple(int(s) for s in spatial_size), str(nn.device()))
        if key not in self._cache:
            self._cache[key] = self.get_embed_host(spatial_size).contiguous().to(nn.device())
        return self._cache[key]
