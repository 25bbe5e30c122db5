"""MI355X-native build of SOC's per-clip inference hot path.

Public boundary (mirrors the reference): build_model(args) -> (model, criterion, postprocessor),
SOC.forward(samples, valid_indices, text_queries, targets), NestedTensor helpers,
MultiScaleDeformableAttention-compatible ms_deform_attn_forward, post-processors."""
from .nested_tensor import NestedTensor, inverse_sigmoid, nested_tensor_from_videos_list  # noqa: F401
from .soc import SOC, build, build_model  # noqa: F401
from .config import default_args  # noqa: F401

__all__ = ["NestedTensor", "nested_tensor_from_videos_list", "inverse_sigmoid", "SOC", "build",
           "build_model", "default_args"]
