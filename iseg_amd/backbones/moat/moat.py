"""backbones/moat/moat.py of the reference: MOAT (:44-242) and moat0 .. moat4 (:245-299)."""
import torch

from ... import functional as F
from ...layers.base_layers import Conv2D
from ...layers.normalizations import normalization
from ...nn import Layer
from .moat_blocks import MBConvBlock, MOATBlock, _INIT

_STRIDE_16_POSITION_EMBEDDING_SIZE = 14
_STRIDE_32_POSITION_EMBEDDING_SIZE = 7
DEFAULT_POS_EMB_SIZE = [None, None, _STRIDE_16_POSITION_EMBEDDING_SIZE, _STRIDE_32_POSITION_EMBEDDING_SIZE]


class MOAT(Layer):
    def __init__(self, stem_size, block_type_list, num_blocks, hidden_size, stage_stride=[2, 2, 2, 2], expansion_rate=4, se_ratio=0.25, head_size=32,
                 window_size=[None, None, None, None], position_embedding_size=DEFAULT_POS_EMB_SIZE, use_checkpointing_for_attention=False,
                 global_attention_at_end_of_moat_stage=False, relative_position_embedding_type="2d_multi_head", ln_epsilon=1e-5, pool_size=2,
                 survival_prob=None, return_endpoints=False, name="moat", trainable=True, **kwargs):
        super().__init__(name=name, trainable=trainable)
        stage_number = len(block_type_list)
        if position_embedding_size is None:      # (:81-83)
            position_embedding_size = [None] * stage_number
            relative_position_embedding_type = None
        if len(num_blocks) != stage_number or len(hidden_size) != stage_number:
            raise ValueError("The lengths of block_type, num_blocks and hidden_size should be the same.")
        self.stem_size, self.block_type, self.num_blocks, self.hidden_size = stem_size, block_type_list, num_blocks, hidden_size
        self.stage_stride, self.expansion_rate, self.se_ratio, self.head_size = stage_stride, expansion_rate, se_ratio, head_size
        self.window_size, self.position_embedding_size = window_size, position_embedding_size
        self.global_attention_at_end_of_moat_stage = global_attention_at_end_of_moat_stage
        self.relative_position_embedding_type, self.ln_epsilon, self.pool_size = relative_position_embedding_type, ln_epsilon, pool_size
        self.survival_prob, self.return_endpoints = survival_prob, return_endpoints

    def _adjust_survival_rate(self, block_id, total_num_blocks):
        if self.survival_prob is None:
            return None
        return 1.0 - (1.0 - self.survival_prob) * block_id / total_num_blocks

    def build(self, input_shape):
        stem = []
        for i in range(len(self.stem_size)):      # (:113-137) Sequential named "stem": conv_i, and norm_i + gelu behind all but the last
            stem.append(Conv2D(self.stem_size[i], 3, strides=2 if i == 0 else 1, padding="same", use_bias=True, kernel_initializer=_INIT,
                               name=f"{self.name}/stem/conv_{i}"))
            if i < len(self.stem_size) - 1:
                stem.append(normalization(name=f"{self.name}/stem/norm_{i}"))
        self._stem = torch.nn.ModuleList(stem)
        self._blocks = torch.nn.ModuleList()
        total = sum(self.num_blocks)
        for stage_id, kind in enumerate(self.block_type):
            stage = torch.nn.ModuleList()
            for local in range(self.num_blocks[stage_id]):
                stride = self.stage_stride[stage_id] if local == 0 else 1
                block_id = sum(self.num_blocks[:stage_id]) + local
                name = f"{self.name}/block_{stage_id:0>2d}_{local:0>2d}"
                window = self.window_size[stage_id]
                if local == self.num_blocks[stage_id] - 1 and kind == "moat" and self.global_attention_at_end_of_moat_stage:
                    window = None
                if kind == "mbconv":
                    stage.append(MBConvBlock(hidden_size=self.hidden_size[stage_id], expansion_rate=self.expansion_rate, se_ratio=self.se_ratio,
                                             block_stride=stride, pool_size=self.pool_size, survival_prob=self._adjust_survival_rate(block_id, total),
                                             name=name))
                elif kind == "moat":      # (:200: the UNADJUSTED survival probability goes to the MOAT blocks)
                    stage.append(MOATBlock(hidden_size=self.hidden_size[stage_id], expansion_rate=self.expansion_rate, block_stride=stride,
                                           pool_size=self.pool_size, survival_prob=self.survival_prob, head_size=self.head_size, window_size=window,
                                           relative_position_embedding_type=self.relative_position_embedding_type,
                                           position_embedding_size=self.position_embedding_size[stage_id], ln_epsilon=self.ln_epsilon, name=name))
                else:
                    raise ValueError(f"Unsupported block_type: {kind}")
            self._blocks.append(stage)
        self.built = True

    def call(self, inputs, training=None):
        x = F.cast_input(inputs)
        for layer in self._stem:
            if isinstance(layer, Conv2D):
                x = layer(x)
            else:
                x = F.gelu(layer(x, training=training))
        endpoints = []
        if self.return_endpoints:
            x, keep = F.fork(x, 2)
            endpoints.append(keep)
        for stage in self._blocks:
            for block in stage:
                x = block(x, training=training)
            if self.return_endpoints:
                x, keep = F.fork(x, 2)
                endpoints.append(keep)
        if self.return_endpoints:
            assert len(endpoints) == 5
            return endpoints
        return x


def _moat(stem, blocks, hidden, survival, return_endpoints, use_pos_emb):
    return MOAT(stem_size=stem, block_type_list=["mbconv", "mbconv", "moat", "moat"], num_blocks=blocks, hidden_size=hidden,
                position_embedding_size=DEFAULT_POS_EMB_SIZE if use_pos_emb else None, survival_prob=survival, return_endpoints=return_endpoints)


def moat0(return_endpoints=False, use_pos_emb=True):
    return _moat([64, 64], [2, 3, 7, 2], [96, 192, 384, 768], 0.8, return_endpoints, use_pos_emb)


def moat1(return_endpoints=False, use_pos_emb=True):
    return _moat([64, 64], [2, 6, 14, 2], [96, 192, 384, 768], 0.7, return_endpoints, use_pos_emb)


def moat2(return_endpoints=False, use_pos_emb=True):
    return _moat([128, 128], [2, 6, 14, 2], [128, 256, 512, 1024], 0.7, return_endpoints, use_pos_emb)


def moat3(return_endpoints=False, use_pos_emb=True):
    return _moat([160, 160], [2, 12, 28, 2], [160, 320, 640, 1280], 0.4, return_endpoints, use_pos_emb)


def moat4(return_endpoints=False, use_pos_emb=True):
    return _moat([256, 256], [2, 12, 28, 2], [256, 512, 1024, 2048], 0.3, return_endpoints, use_pos_emb)
