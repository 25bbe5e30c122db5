"""The reference's native plugin under its own module name.

The reference binds its CUDA extension as ``import MultiScaleDeformableAttention as MSDA``
(models/ops/functions/ms_deform_attn_func.py:18; built by models/ops/setup.py:59-66 from
src/vision.cpp:13-16, which exports exactly two functions).  With this repository's root on
``sys.path`` that import resolves here, and both entry points run the gfx950 kernels behind the C ABI
(include/soc_hip.h: soc_msda_fwd_f32/f64, soc_msda_bwd_f32/f64).  Same argument order, dtype rule
(float32 / float64), contiguity and im2col_step errors as the extension.  No CPU path: CPU tensors raise,
as the reference's CPU entry points do (src/cpu/ms_deform_attn_cpu.cpp:26,40).
"""
from neurips2023_soc_amd.ms_deform_attn import ms_deform_attn_backward, ms_deform_attn_forward

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward"]
