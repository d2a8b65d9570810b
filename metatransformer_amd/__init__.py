"""metatransformer_amd -- MI355X-native Meta-Transformer encoder hot path.

Drop-in for the reference's `timm.models.vision_transformer.Block` stack and the Data2Seq tokenizers that
feed it (see encoder.py / data2seq.py); all compute runs in libmetaenc.so (hand-written HIP for gfx950)
behind the C ABI of include/metaenc.h.
"""
from ._capi import MetaEncError, load as load_library  # noqa: F401
from .encoder import (Attention, Block, Mlp, build_encoder, set_fp32_mode, convert_video_state_dict, encoder_flops_per_sample,  # noqa: F401
                      encoder_forward_inference, resize_pos_embed, to_video_state_dict)
from .heads import (ClassifierHead, ClsHead, PointPatchEmbed, furthest_point_sample, knn_indices,  # noqa: F401
                    load_encoder_checkpoint, pack_encoder, pool_tokens, save_encoder_checkpoint)
from .data2seq import (AcousticPatchEmbed, Data2Seq, DataEmbedding, PatchEmbed, VideoPatchEmbed,  # noqa: F401
                       sinusoid_table, video_sinusoid_table)

__all__ = ["Block", "Attention", "Mlp", "build_encoder", "set_fp32_mode", "encoder_flops_per_sample", "encoder_forward_inference", "Data2Seq", "PatchEmbed",
           "AcousticPatchEmbed", "VideoPatchEmbed", "DataEmbedding", "MetaEncError", "load_library",
           "convert_video_state_dict", "to_video_state_dict", "resize_pos_embed", "ClassifierHead", "ClsHead", "PointPatchEmbed",
           "furthest_point_sample", "knn_indices", "pool_tokens", "load_encoder_checkpoint", "save_encoder_checkpoint", "pack_encoder"]
