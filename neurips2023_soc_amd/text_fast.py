"""The text side of SOC.forward_text on the GPU: HuggingFace RobertaModel with its encoder layers re-expressed on this
package's small-row kernels.

The reference runs `RobertaModel` per forward (models/soc.py:167-181); north_star keeps RoBERTa in PyTorch.  On ~10 word
rows the stock module is ~170 short launches per clip (72 library GEMMs of 10-17 us on a few workgroups each, separate
bias-free adds, LayerNorms, GELU) that run on the side branch beside Video-Swin and cost it 0.27 ms per clip
(tools/experiments/text_ablation.py).  Here a layer is 7 launches:

    K7 (q | k | v as three segments of one launch)  ->  F.scaled_dot_product_attention (what HF's "sdpa" path calls)
    ->  K7 (attention.output.dense)  ->  K5 (residual + LayerNorm)
    ->  K7 (intermediate.dense with the exact-erf GELU in its epilogue)  ->  K7 (output.dense)  ->  K5

Parameters, state_dict keys and the module tree are HuggingFace's; the layer instances are re-classed to a subclass that
overrides `forward`, which falls back to the original for anything it does not cover (CPU, training, decoder / cross-attention, caches, attention
probabilities requested, another activation).  Embeddings, mask construction and the pooler stay HuggingFace code.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import fused, hot_ops


class RoutedLinear(nn.Linear):
    """nn.Linear whose forward goes through fused.linear (K7 for few rows, the library GEMM otherwise).  Instances are
    made by re-classing existing nn.Linear modules, so parameters and state_dict keys are untouched and copies / pickles of
    the model keep working (a method bound to an instance would keep pointing at the original module's weights)."""

    def forward(self, x):
        # the K7 path has no autograd: anything that may need a gradient (fine-tuning with freeze_text_encoder=False)
        # goes through F.linear
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)
        if x.is_cuda and x.dtype == torch.float32 and not needs_grad:
            return fused.linear(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


FAST_LAYER_CALLS = 0     # encoder-layer forwards that took the 7-launch form (tests assert that it is not dead code)


def _covered(layer, hidden_states, args, kwargs) -> bool:
    # HuggingFace passes some of the optional arguments POSITIONALLY (transformers 5.x: RobertaEncoder.forward hands
    # `encoder_hidden_states` over as the third positional argument; 4.x: head_mask, ..., output_attentions), in an order
    # that differs between releases.  Whatever they are, a None / False there requests nothing.
    if any(a is not None and a is not False for a in args):
        return False
    if layer.training or not hidden_states.is_cuda or hidden_states.dtype != torch.float32:
        return False
    if torch.is_grad_enabled() and any(p.requires_grad for p in layer.attention.self.query.parameters()):
        return False
    if getattr(layer, "is_decoder", False) or getattr(layer, "add_cross_attention", False):
        return False
    if kwargs.get("output_attentions") or kwargs.get("head_mask") is not None:
        return False
    if any(kwargs.get(k) is not None for k in ("encoder_hidden_states", "encoder_attention_mask", "past_key_values",
                                               "past_key_value")):
        return False
    return hidden_states.dim() == 3 and fused.is_small(hidden_states)


def _additive_mask(mask, dtype):
    """HF passes the SAME boolean [B,1,L,L] mask object to all 12 layers and SDPA turns a boolean mask into an additive
    one inside every call (two fills, a where and a copy: 48 tiny launches per clip).  Converted once per forward here;
    the result rides on the mask object itself, so it cannot outlive it (graph captures make new masks)."""
    if mask is None or mask.dtype != torch.bool:
        return mask
    cached = getattr(mask, "_soc_additive", None)
    if cached is None or cached[0] != (mask._version, dtype):
        add = torch.zeros(mask.shape, dtype=dtype, device=mask.device).masked_fill_(~mask, float("-inf"))
        cached = ((mask._version, dtype), add)
        mask._soc_additive = cached
    return cached[1]


def _layer_forward(self, hidden_states, attention_mask=None, *args, **kwargs):
    if not _covered(self, hidden_states, args, kwargs):
        return self._soc_orig_forward(hidden_states, attention_mask, *args, **kwargs)       # the parent class's forward
    global FAST_LAYER_CALLS
    FAST_LAYER_CALLS += 1
    att, att_out, inter, out = self.attention.self, self.attention.output, self.intermediate, self.output
    B, L, E = hidden_states.shape
    nh = att.num_attention_heads
    q, k, v = fused.linear_multi(hidden_states, [(att.query.weight, att.query.bias, False),
                                                 (att.key.weight, att.key.bias, False),
                                                 (att.value.weight, att.value.bias, False)])
    amask = _additive_mask(attention_mask, q.dtype)
    if hot_ops.small_attention_supported(q, nh, amask):
        # K25: one workgroup per (batch, head) on the token-major projections (no transposes, no Triton-compiled kernel)
        ctx = hot_ops.small_attention(q, k, v, nh, amask, getattr(att, "scaling", None))
    else:
        q, k, v = (t.view(B, L, nh, E // nh).transpose(1, 2) for t in (q, k, v))
        ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=amask, scale=getattr(att, "scaling", None))
        ctx = ctx.transpose(1, 2).reshape(B, L, E)
    h = hot_ops.add_layernorm(hidden_states, fused.linear(ctx, att_out.dense.weight, att_out.dense.bias),
                              att_out.LayerNorm.weight, att_out.LayerNorm.bias, att_out.LayerNorm.eps, return_sum=False)[1]
    mid = hot_ops.linear_small(h, inter.dense.weight, inter.dense.bias, None, "gelu")
    y = hot_ops.add_layernorm(h, fused.linear(mid, out.dense.weight, out.dense.bias), out.LayerNorm.weight,
                              out.LayerNorm.bias, out.LayerNorm.eps, return_sum=False)[1]
    return y if self._soc_returns_tensor else (y,)


_FAST_CLASSES = {}


def _fast_layer_class(cls):
    """Subclass of a HuggingFace encoder-layer class whose forward is _layer_forward (the original stays reachable as
    _soc_orig_forward).  One per original class, registered in this module so that pickling by reference works."""
    if cls not in _FAST_CLASSES:
        import transformers
        name = "Fast" + cls.__name__
        sub = type(name, (cls,), {"forward": _layer_forward, "_soc_orig_forward": cls.forward,
                                  "_soc_returns_tensor": int(transformers.__version__.split(".")[0]) >= 5,   # 4.x: tuple
                                  "__module__": __name__})
        globals()[name] = sub
        _FAST_CLASSES[cls] = sub
    return _FAST_CLASSES[cls]


def accelerate_text_encoder(text_encoder: nn.Module) -> int:
    """Re-class every RobertaLayer-shaped module of `text_encoder` to its fast subclass and its remaining nn.Linear layers
    (pooler) to RoutedLinear.  Idempotent.  Returns the number of encoder layers on the fast forward (0: unknown module
    layout, nothing changed except the Linear routing)."""
    n = 0
    for m in text_encoder.modules():
        if type(m) is nn.Linear and m.in_features % 16 == 0:
            m.__class__ = RoutedLinear
    for m in text_encoder.modules():
        if type(m) in _FAST_CLASSES.values():
            n += 1
            continue
        try:
            att, inter, out = m.attention.self, m.intermediate, m.output
            shaped = (all(isinstance(x, nn.Linear) for x in (att.query, att.key, att.value, m.attention.output.dense,
                                                             inter.dense, out.dense))
                      and type(m.attention.output.LayerNorm) is nn.LayerNorm and type(out.LayerNorm) is nn.LayerNorm)
        except AttributeError:
            continue
        act = getattr(inter, "intermediate_act_fn", None)
        exact_gelu = type(act).__name__ == "GELUActivation" or act is F.gelu or isinstance(act, nn.GELU)
        if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") != "none":
            exact_gelu = False
        if not (shaped and exact_gelu) or getattr(att, "position_embedding_type", "absolute") != "absolute":
            continue
        m.__class__ = _fast_layer_class(type(m))
        n += 1
    return n
