"""backbones/vit.py of the reference: resize_pos_embed (:19-63), MLPBlock (:66-113), TransformerBlock (:116-183),
VisionTransformer (:186-323), ViT16B / ViT16L (:326-354).  Tokens are [B, T, C]; attention is the packed-qkv operator of
functional.attention_packed behind keras-MultiHeadAttention-shaped weights."""
import numpy as np
import torch

from .. import functional as F
from .. import nn
from ..layers.base_layers import Dense, Dropout, LayerNormalization
from ..layers.common_layers import PatchEmbed
from ..layers.keras_mha import MultiHeadAttention
from ..layers.model_builder import get_training_value
from ..nn import Layer
from ..utils.bicubic import bicubic_matrix

_TN002 = ("truncated_normal", 0.02)


class MLPBlock(Layer):
    def __init__(self, filters, dropout_rate=0.0, activation="gelu", name=None):
        super().__init__(name=name)
        self.filters, self.dropout_rate, self.activation = filters, dropout_rate, activation

    def build(self, input_shape):
        self.dense0 = Dense(self.filters, activation=self.activation, kernel_initializer=_TN002, name=f"{self.name}/dense0")
        self.dense0_dropout = Dropout(self.dropout_rate)
        self.dense1 = Dense(int(input_shape[-1]), kernel_initializer=_TN002, name=f"{self.name}/dense1")
        self.dense1_dropout = Dropout(self.dropout_rate, name="dense1_dropout")
        self.built = True

    def fusable(self, training):
        return (self.activation == "gelu" and self.built and self.dense0.built and self.dense1.built
                and (self.dropout_rate == 0.0 or not training))

    def call(self, inputs, training=None, residual=None, drop_path_mask=None):
        """residual / drop_path_mask (fusable(training) only): residual + factor[sample] * mlp(inputs) from the second product's epilogue"""
        if self.fusable(training):
            return F.mlp_gelu(inputs, self.dense0.kernel, self.dense0.bias, self.dense1.kernel, self.dense1.bias, residual=residual,
                              drop_path_mask=drop_path_mask)      # one tape node
        assert residual is None and drop_path_mask is None
        x = self.dense0(inputs)
        x = self.dense0_dropout(x, training=training)
        x = self.dense1(x)
        return self.dense1_dropout(x, training=training)


class TransformerBlock(Layer):
    def __init__(self, mlp_filters=4096, num_heads=16, dropout_rate=0.1, drop_path_rate=0.0, name=None):
        super().__init__(name=name)
        self.num_head, self.mlp_filters = num_heads, mlp_filters
        self.dropout_rate, self.drop_path_rate = dropout_rate, float(drop_path_rate)
        self.drop_path_masks = None      # parity tests may inject the two per-sample factor vectors

    def build(self, input_shape):
        channels = int(input_shape[-1])
        self.attention_norm = LayerNormalization(epsilon=1e-6, name=f"{self.name}/ln1")
        self.attention = MultiHeadAttention(num_heads=self.num_head, key_dim=channels // self.num_head, dropout=self.dropout_rate,
                                            name=f"{self.name}/attn")
        self.mlp_norm = LayerNormalization(epsilon=1e-6, name=f"{self.name}/ln2")
        self.mlp = MLPBlock(self.mlp_filters, self.dropout_rate, name=f"{self.name}/ffn")
        self.built = True

    def call(self, inputs, training=None):
        training = get_training_value(training)
        masks = self.drop_path_masks or (None, None)
        branch, skip = F.fork(inputs, 2)      # residual forks: their gradients are summed by our own kernel, not by autograd
        x = self.attention_norm(branch)
        x = self.attention(x, x, training=training)
        if self.drop_path_rate != 0.0 and training:
            x = F.drop_path(x, self.drop_path_rate, training, mask=masks[0])
        x, identity = F.fork(F.add(x, skip), 2)
        x = self.mlp_norm(x)
        if self.mlp.fusable(training):      # identity + drop_path(mlp(.)) out of the second product's epilogue
            mask = None
            if self.drop_path_rate != 0.0 and training:
                mask = masks[1] if masks[1] is not None else F.drop_path_factors(x.shape[0], 1.0 - self.drop_path_rate, x.device)
            return self.mlp(x, training=training, residual=identity, drop_path_mask=mask)
        x = self.mlp(x, training=training)
        if self.drop_path_rate != 0.0 and training:
            x = F.drop_path(x, self.drop_path_rate, training, mask=masks[1])
        return F.add(x, identity)


class VisionTransformer(Layer):
    def __init__(self, patch_size, num_layer, num_head, filters=768, mlp_filters=4096, dropout_rate=0.0, drop_path_rate=0.1,
                 use_class_token=True, pretrain_size=224, return_endpoints=False, name=None):
        super().__init__(name=name)
        self.patch_size, self.num_layer, self.num_head = patch_size, num_layer, num_head
        self.filters, self.mlp_filters = filters, mlp_filters
        self.dropout_rate, self.drop_path_rate = dropout_rate, drop_path_rate
        self.use_class_token, self.pretrain_size, self.return_endpoints = use_class_token, pretrain_size, return_endpoints
        self._resize_cache = {}

    def build(self, input_shape):
        self.patch_encoder = PatchEmbed(patch_size=(self.patch_size, self.patch_size), embed_filters=self.filters,
                                        name=f"{self.name}/patch_embed")
        axis = self.pretrain_size // self.patch_size
        self.num_patches_axis = axis
        self.num_patches = axis ** 2
        self.extra_patches = 0
        if self.use_class_token:
            self.class_token = self.add_weight("class_token", (1, 1, self.filters), "zeros")
            self.extra_patches = 1
        self.position_embedding = self.add_weight("pos_embed", (1, self.num_patches + self.extra_patches, self.filters), _TN002)
        self.position_embedding_dropout = Dropout(self.dropout_rate, name="position_embedding_dropout")
        rates = np.linspace(0.0, self.drop_path_rate, self.num_layer)
        self.blocks = torch.nn.ModuleList([
            TransformerBlock(self.mlp_filters, num_heads=self.num_head, dropout_rate=self.dropout_rate, drop_path_rate=float(rates[i]),
                             name=f"{self.name}/layers/{i}") for i in range(self.num_layer)])
        self.built = True

    def _resize_matrices(self, height, width):
        key = (height, width)
        if key not in self._resize_cache:
            g = self.num_patches_axis
            self._resize_cache[key] = (torch.from_numpy(bicubic_matrix(height, g)).to(nn.device()),
                                       torch.from_numpy(bicubic_matrix(width, g)).to(nn.device()))
        return self._resize_cache[key]

    def call(self, inputs, training=None):
        x = F.cast_input(inputs)
        x = self.patch_encoder(x)
        batch_size, height, width, channels = x.shape
        x = x.reshape(batch_size, height * width, channels)            # flatten_hw
        if self.use_class_token:
            x = F.prepend_token(x, self.class_token)
        wy, wx = self._resize_matrices(height, width)
        pos = F.resize_pos_embed(self.position_embedding, wy, wx, self.extra_patches, x.dtype)
        x = F.add_batch_broadcast(x, pos)
        x = self.position_embedding_dropout(x, training=training)
        for blk in self.blocks:
            x = blk(x, training=training)
        if self.use_class_token:
            x = F.drop_tokens(x, self.extra_patches)
        x = x.reshape(batch_size, height, width, channels)
        return [x] if self.return_endpoints else x


def ViT16L(return_endpoints=False):
    return VisionTransformer(patch_size=16, num_layer=24, num_head=16, filters=1024, mlp_filters=4096, pretrain_size=384,
                             use_class_token=True, return_endpoints=return_endpoints, name="ViT-L_16")


def ViT16B(return_endpoints=False):
    return VisionTransformer(patch_size=16, num_layer=12, num_head=12, filters=768, mlp_filters=3072, pretrain_size=384,
                             use_class_token=True, return_endpoints=return_endpoints, name="ViT-B_16")


def ViT16S(return_endpoints=False):
    return VisionTransformer(patch_size=16, num_layer=12, num_head=6, filters=384, mlp_filters=1536, pretrain_size=384,
                             use_class_token=True, return_endpoints=return_endpoints, name="ViT-S_16")
