"""Host-side mirror of the reference's module interface.  TARGETS maps the `_target_` strings used by the reference's
Hydra YAMLs (sam2_train/sam2_hiera_*.yaml) to the drop-in classes, so an unmodified YAML instantiates this package."""
from .encoder import FpnNeck, Hiera, ImageEncoder, MLP, MultiScaleAttention, MultiScaleBlock, PatchEmbed
from .memory import (Attention, CXBlock, Fuser, LayerNorm2d, MaskDownSampler, MemoryAttention, MemoryAttentionLayer,
                     MemoryEncoder, RoPEAttention)
from .position import PositionEmbeddingRandom, PositionEmbeddingSine
from .sam2_base import SAM2Base
from .sam_heads import MaskDecoder, PromptEncoder, TwoWayAttentionBlock, TwoWayTransformer

TARGETS = {
    "sam2_train.modeling.sam2_base.SAM2Base": SAM2Base,
    "sam2_train.modeling.backbones.image_encoder.ImageEncoder": ImageEncoder,
    "sam2_train.modeling.backbones.image_encoder.FpnNeck": FpnNeck,
    "sam2_train.modeling.backbones.hieradet.Hiera": Hiera,
    "sam2_train.modeling.position_encoding.PositionEmbeddingSine": PositionEmbeddingSine,
    "sam2_train.modeling.memory_attention.MemoryAttention": MemoryAttention,
    "sam2_train.modeling.memory_attention.MemoryAttentionLayer": MemoryAttentionLayer,
    "sam2_train.modeling.sam.transformer.RoPEAttention": RoPEAttention,
    "sam2_train.modeling.memory_encoder.MemoryEncoder": MemoryEncoder,
    "sam2_train.modeling.memory_encoder.MaskDownSampler": MaskDownSampler,
    "sam2_train.modeling.memory_encoder.Fuser": Fuser,
    "sam2_train.modeling.memory_encoder.CXBlock": CXBlock,
}
