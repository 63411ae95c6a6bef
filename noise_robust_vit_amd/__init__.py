"""MI355X-native ViT encoder-block hot path: drop-in for noise-robust-vit's SimpleViT / VisionTransformer modules.

    from noise_robust_vit_amd import SimpleViT            # was: from vit_pytorch_robust import SimpleViT
    from noise_robust_vit_amd.vit import vit_b_16         # was: from vit_pytorch_robust.vit import vit_b_16

The arithmetic lives in hand-written gfx950 HIP kernels behind a C ABI (include/nrv.h, lib/libnrv_hip.so).
Importing this package does not need a GPU; running a forward does, and fails loudly otherwise.
"""
from .simple_vit import Attention, FeedForward, SimpleViT, SinkhornAttention, Transformer  # noqa: F401
from .vit import VisionTransformer, vit_b_16, vit_b_32, vit_l_16, vit_l_32, vit_s_16  # noqa: F401



def invalidate_weight_cache() -> None:
    """Drop the bf16 images of the weights (encoder.WeightCache).  They are keyed on each parameter's autograd version
    counter; call this after updating parameters by any means that does not bump it (raw-pointer kernels, some fused
    multi-tensor optimizers).  `train.Trainer` and `optim.FusedAdamW` do it themselves."""
    from .encoder import WEIGHTS
    WEIGHTS.clear()


__all__ = ["invalidate_weight_cache", "SimpleViT", "Attention", "FeedForward", "Transformer", "SinkhornAttention",
           "VisionTransformer", "vit_s_16", "vit_b_16", "vit_b_32", "vit_l_16", "vit_l_32"]
