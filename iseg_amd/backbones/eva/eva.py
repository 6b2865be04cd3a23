"""backbones/eva/eva.py of the reference: Eva (:22-312) and the EVA02 constructors (:316-467).  Tokens are [B, 1 + H W, C]; the rotary table of the
token grid comes from RotaryEmbeddingCat once per (H, W) and is applied inside every block's attention by the C ABI's iseg_qkv_rope."""
import numpy as np
import torch

from ... import functional as F
from ... import nn
from ...layers.base_layers import Dropout
from ...layers.common_layers import PatchEmbed, to_2d_tuple
from ...nn import Layer
from .block import EvaBlock
from .rotar_embedding_cat import RotaryEmbeddingCat


def bilinear_matrix(out_size, in_size):
    """tf.image.resize(method="bilinear", half-pixel centres, no antialias) along one axis as a [out, in] matrix"""
    m = np.zeros((out_size, in_size), dtype=np.float32)
    scale = in_size / out_size
    for o in range(out_size):
        src = (o + 0.5) * scale - 0.5
        f = np.floor(src)
        lo, hi = int(max(f, 0)), int(min(np.ceil(src), in_size - 1))
        t = np.float32(src - f)
        m[o, lo] += 1.0 - t
        m[o, hi] += t
    return m


class Eva(Layer):
    def __init__(self, pretrain_img_size=224, pretrain_patch_size=14, patch_size=16, embed_filters=768, depth=12, num_heads=12, qkv_bias=True,
                 qkv_fused=True, mlp_ratio=4.0, swiglu_mlp=False, scale_mlp=False, scale_attention_inner=False, droppout_rate=0.0,
                 pos_droppout_rate=0.0, attention_dropout_rate=0.0, projection_dropout_rate=0.0, drop_path_rate=0.0, init_values=None,
                 use_class_token=True, use_abs_pos_emb=True, use_rot_pos_emb=True, use_post_norm=False, dynamic_img_size=True, ref_feat_shape=None,
                 patch_padding="valid", return_endpoints=False, trainable=True, name=None):
        super().__init__(name=name, trainable=trainable)
        self.pretrain_img_size, self.pretrain_patch_size, self.patch_size = pretrain_img_size, pretrain_patch_size, patch_size
        self.embed_filters, self.depth, self.num_heads = embed_filters, depth, num_heads
        self.qkv_bias, self.qkv_fused, self.mlp_ratio, self.swiglu_mlp, self.scale_mlp = qkv_bias, qkv_fused, mlp_ratio, swiglu_mlp, scale_mlp
        self.scale_attention_inner = scale_attention_inner
        self.droppout_rate, self.pos_droppout_rate = droppout_rate, pos_droppout_rate
        self.attention_dropout_rate, self.projection_dropout_rate, self.drop_path_rate = attention_dropout_rate, projection_dropout_rate, drop_path_rate
        self.init_values, self.use_class_token, self.use_abs_pos_emb, self.use_rot_pos_emb = init_values, use_class_token, use_abs_pos_emb, use_rot_pos_emb
        self.use_post_norm, self.dynamic_img_size, self.ref_feat_shape = use_post_norm, dynamic_img_size, ref_feat_shape
        self.patch_padding, self.return_endpoints = patch_padding, return_endpoints
        self._resize_cache = {}

    def build(self, input_shape):
        input_height, input_width = int(input_shape[1]), int(input_shape[2])
        self.patch_embed = PatchEmbed(patch_size=self.patch_size, weights_patch_size=self.pretrain_patch_size, embed_filters=self.embed_filters,
                                      padding=self.patch_padding, name=f"{self.name}/patch_embed")
        self.grid_size = [input_height // self.pretrain_patch_size, input_width // self.pretrain_patch_size]
        num_patches = self.grid_size[0] * self.grid_size[1]
        self.num_prefix_tokens = 1 if self.use_class_token else 0
        self.class_token = self.add_weight("class_token", (1, 1, self.embed_filters), "zeros") if self.use_class_token else None
        self.position_embedding = (self.add_weight("pos_embed", (1, num_patches + self.num_prefix_tokens, self.embed_filters), "zeros")
                                   if self.use_abs_pos_emb else None)
        if self.position_embedding is not None:
            # (:131-165) the reference wraps pos_embed.assign: the FIRST value assigned (the pretrained table, laid out on the pretrain grid
            # pretrain_img_size // pretrain_patch_size) is resampled bicubically to the build grid, prefix tokens passed through
            # (utils/common.py:206-262 resample_absolute_position_embedding); later assignments are taken as they come
            ph, pw = to_2d_tuple(self.pretrain_img_size)
            self._pretrain_grid = (int(ph) // self.pretrain_patch_size, int(pw) // self.pretrain_patch_size)
            self._pos_embed_assigned = False
            self.position_embedding.iseg_assign_hook = self._resample_on_first_assign
        self.pos_droppout = Dropout(self.pos_droppout_rate, name="pos_droppout")
        self.rope = None
        if self.use_rot_pos_emb:
            ref = to_2d_tuple(self.ref_feat_shape) if self.ref_feat_shape is not None else None
            self.rope = RotaryEmbeddingCat(filters=self.embed_filters // self.num_heads, in_pixels=False,
                                           feat_shape=None if self.dynamic_img_size else list(self.grid_size), ref_feat_shape=ref)
        dpr = np.linspace(0.0, self.drop_path_rate, self.depth)
        self.blocks = torch.nn.ModuleList([
            EvaBlock(num_heads=self.num_heads, qkv_bias=self.qkv_bias, qkv_fused=self.qkv_fused, mlp_ratio=self.mlp_ratio, swiglu_mlp=self.swiglu_mlp,
                     scale_mlp=self.scale_mlp, scale_attention_inner=self.scale_attention_inner, attention_dropout_rate=self.attention_dropout_rate,
                     projection_dropout_rate=self.projection_dropout_rate, drop_path_rate=float(dpr[i]), init_values=self.init_values,
                     use_post_norm=self.use_post_norm, class_token_size=self.num_prefix_tokens, name=f"{self.name}/blocks/{i}")
            for i in range(self.depth)])
        self.built = True

    def _resample_on_first_assign(self, value):
        """value [1, prefix + ph * pw, C] (the pretrain grid) -> [1, prefix + gh * gw, C] by tf.image.resize(method="bicubic") arithmetic
        (utils/bicubic.bicubic_matrix: half-pixel centres, Keys a = -0.5 table, no antialias); a value already on the build grid passes"""
        from ...utils.bicubic import bicubic_matrix

        value = np.asarray(value, dtype=np.float32)
        if self._pos_embed_assigned:
            return value
        self._pos_embed_assigned = True
        n_prefix = self.num_prefix_tokens
        (ph, pw), (gh, gw) = self._pretrain_grid, self.grid_size
        tokens = value.shape[-2] - n_prefix
        if tokens == gh * gw and (ph, pw) == (gh, gw):
            return value
        if tokens != ph * pw:
            if tokens == gh * gw:      # a table already on the build grid (a checkpoint of this model)
                return value
            raise ValueError(f"{self.name}/pos_embed: {tokens} stored position tokens, expected the pretrain grid {ph} x {pw} "
                             f"(pretrain_img_size {self.pretrain_img_size} // patch {self.pretrain_patch_size}) or the build grid {gh} x {gw}")
        value = value.reshape(1, n_prefix + tokens, -1)
        spatial = value[0, n_prefix:].reshape(ph, pw, -1)
        wy, wx = bicubic_matrix(gh, ph), bicubic_matrix(gw, pw)
        out = np.einsum("oh,hwc->owc", wy, spatial)
        out = np.einsum("pw,owc->opc", wx, out).reshape(gh * gw, -1).astype(np.float32)
        return np.concatenate([value[0, :n_prefix], out], axis=0)[None]

    def _resize_matrices(self, height, width):
        key = (height, width)
        if key not in self._resize_cache:
            self._resize_cache[key] = (torch.from_numpy(bilinear_matrix(height, self.grid_size[0])).to(nn.device()),
                                       torch.from_numpy(bilinear_matrix(width, self.grid_size[1])).to(nn.device()))
        return self._resize_cache[key]

    def _pos_embed(self, x, training=None):
        b, h, w, c = x.shape
        x = x.reshape(b, h * w, c)
        if self.use_class_token:
            x = F.prepend_token(x, self.class_token)
        if self.position_embedding is not None:
            # (:245-254) dynamic_img_size: the embedding of the build-time grid is resampled bilinearly to this call's grid (the identity map when they
            # agree); otherwise the token counts must agree
            if not self.dynamic_img_size and [h, w] != list(self.grid_size):
                raise ValueError(f"Eva(dynamic_img_size=False): {h} x {w} tokens, the position embedding was built for {self.grid_size}")
            wy, wx = self._resize_matrices(h, w)
            pos = F.resize_pos_embed(self.position_embedding, wy, wx, self.num_prefix_tokens, x.dtype)
            x = F.add_batch_broadcast(x, pos)
        return self.pos_droppout(x, training=training)

    def call(self, inputs, training=None):
        x = F.cast_input(inputs)
        x = self.patch_embed(x)
        patch_embedding = x
        b, h, w, c = x.shape
        rope = self.rope([h, w]) if self.rope is not None and not nn.dry_run() else None
        if self.return_endpoints:
            x, patch_embedding = F.fork(x, 2)
        x = self._pos_embed(x, training=training)
        endpoints = []
        for blk in self.blocks:
            x = blk(x, rope=rope, training=training)
            if self.return_endpoints:
                x, tap = F.fork(x, 2)
                endpoints.append(F.drop_tokens(tap, self.num_prefix_tokens).reshape(b, h, w, c))
        if self.return_endpoints:
            class_token = x[:, :1, :] if self.use_class_token else None
            return [class_token, patch_embedding] + endpoints
        return x


def _eva02_large(pretrain_img_size, patch, name, return_endpoints):
    return Eva(pretrain_img_size=pretrain_img_size, pretrain_patch_size=patch, patch_size=patch, embed_filters=1024, depth=24, num_heads=16,
               qkv_fused=False, mlp_ratio=4 * 2 / 3, swiglu_mlp=True, scale_mlp=True, scale_attention_inner=False, drop_path_rate=0.3, init_values=None,
               use_class_token=True, use_abs_pos_emb=True, use_rot_pos_emb=True, use_post_norm=False, return_endpoints=return_endpoints, name=name)


def EVA02_large_patch14_448(return_endpoints=False):
    return _eva02_large(448, 14, "eva02_large_patch14_448", return_endpoints)


def EVA02_large_patch14_224(return_endpoints=False):
    return _eva02_large(224, 14, "eva02_large_patch14_224", return_endpoints)


def EVA02_large_patch16_224(return_endpoints=False):
    return _eva02_large(224, 16, "eva02_large_patch16_224", return_endpoints)


def EVA02_large_patch16_512_COCO(return_endpoints=False):
    return _eva02_large(512, 16, "eva02_large_patch16_512_coco", return_endpoints)


def EVA02_large_patch16_512_MV(return_endpoints=False):
    return _eva02_large((512, 1024), 16, "eva02_large_patch16_512_coco", return_endpoints)      # (the reference reuses the COCO name, :431)


def EVA02_tiny_patch_14_336(return_endpoints=False):
    return Eva(pretrain_img_size=336, pretrain_patch_size=14, patch_size=14, embed_filters=192, depth=12, num_heads=3, qkv_fused=True,
               mlp_ratio=4 * 2 / 3, swiglu_mlp=True, scale_mlp=False, scale_attention_inner=False, drop_path_rate=0.0, init_values=None,
               use_class_token=True, use_abs_pos_emb=True, use_rot_pos_emb=True, use_post_norm=False, return_endpoints=return_endpoints,
               name="eva02_tiny_patch_14_336")
