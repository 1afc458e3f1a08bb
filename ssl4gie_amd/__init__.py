"""ssl4gie_amd — MI355X-native engine for the SSL4GIE data-parallel hot path.

Public surface mirrors the reference's own:
    ssl4gie_amd.Models.models           ViT_from_MAE, VisionTransformer_from_Any, ...
    ssl4gie_amd.Models.mae.models_mae   MaskedAutoencoderViT, mae_vit_base_patch16, ...
    ssl4gie_amd.utils                   get_MAE_backbone, ... factories (reference utils.py)
    ssl4gie_amd.parallel                one-process-per-GPU data parallelism over RCCL
The compute path is libssl4gie_hip.so (include/ssl4gie_hip.h); importing this package does not
need a GPU, running a model does (there is no CPU fallback).
"""
__version__ = "0.1.0"
