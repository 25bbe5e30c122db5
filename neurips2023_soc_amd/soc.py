"""SOC model, MI355X build: same constructor arguments, forward signature, output dict and
state_dict key names as the reference (models/soc.py:31-394, build :624-660) so it drops into
infer_refytb.py:133-214 / infer_davis.py unchanged -- but inference-only, with the four hot ops
in hand-written HIP (hot_ops.py -> libsoc_hip.so) and the rest as fp32 library GEMMs/convs.

Differences that do not change results:
  * eval returns the decoder-level-0 heads (reference zip() quirk, soc.py:375-394 + voc.py:274);
    class/box/mask heads of levels 1-2 are never computed here, decoder layers 1-2 still run
    because VOC consumes hs[-1];
  * `lvf` is evaluated only on the last fused level -- the only one the reference reads (:304);
  * the dynamic mask head never materialises the [1, T*Q*10, h, w] tensor (K4).
"""
from __future__ import annotations

import copy
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import fused, hot_ops
from .deformable_transformer import build_deforamble_transformer
from .nested_tensor import NestedTensor
from .position_encoding import PositionEmbeddingSine1D
from .postprocessing import build_postprocessors
from .spatial_decoder import FPNSpatialDecoder
from .text_fast import accelerate_text_encoder
from .video_swin import build_video_swin_backbone, resize_pad_mask
from .vla import MMF
from .voc import VOC


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        ws = [layer.weight for layer in self.layers]
        if hot_ops.row_mlp_supported(x, ws):          # K16: the whole MLP of a few query rows in one launch
            return hot_ops.row_mlp(x, [(layer.weight, layer.bias) for layer in self.layers])
        for i, layer in enumerate(self.layers):
            x = fused.linear(x, layer.weight, layer.bias, relu=i + 1 < self.num_layers)
        return x


class FeatureResizer(nn.Module):
    """Linear + LayerNorm(eps 1e-12) (+ dropout, identity in eval) -- reference :566-585."""

    def __init__(self, input_feat_size, output_feat_size, dropout, do_ln=True):
        super().__init__()
        self.do_ln = do_ln
        self.fc = nn.Linear(input_feat_size, output_feat_size, bias=True)
        self.layer_norm = nn.LayerNorm(output_feat_size, eps=1e-12)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        x = fused.apply(self.fc, x)
        return self.dropout(self.layer_norm(x) if self.do_ln else x)


def roberta_base_config():
    from transformers import RobertaConfig
    return RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                         intermediate_size=3072, max_position_embeddings=514, type_vocab_size=1,
                         layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)


def _load_text_stack(config):
    """(RobertaModel, tokenizer-or-None).  Real weights when `text_encoder_type` resolves offline;
    a random-init roberta-base only when the caller opts in (synthetic benchmarks / tests)."""
    from transformers import RobertaModel, RobertaTokenizerFast
    name = config.text_encoder_type
    try:
        enc = RobertaModel.from_pretrained(name, local_files_only=True)
        return enc, load_roberta_tokenizer(name)
    except Exception as exc:  # no files / no network
        if not getattr(config, "text_encoder_random_init", False):
            raise RuntimeError(
                f"cannot load text encoder {name!r} offline ({type(exc).__name__}); pass "
                "text_encoder_random_init=True for synthetic-weight runs") from exc
        return RobertaModel(roberta_base_config()), None


def _in_model_mode(method):
    """Run a stage of the forward in the model's own arithmetic (SOC.matmul_mode; hot_ops.use_matmul_mode is thread-local,
    so two models with different modes can run from two threads)."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        with hot_ops.use_matmul_mode(self.matmul_mode):
            return method(self, *args, **kwargs)
    return wrapped


def load_roberta_tokenizer(name_or_dir):
    """`RobertaTokenizerFast.from_pretrained(name_or_dir)` (models/soc.py:104), offline.  A directory that holds a
    `tokenizer.json` is loaded from that file: transformers 5.x's `from_pretrained` rebuilds the backend from vocab /
    merges there and comes back WITHOUT the byte-level pre-tokenizer (spaces vanish, no merge applies) -- so the result is
    checked, and a tokenizer that would split text differently from RoBERTa's is an error, not a silent change of ids."""
    import json
    from transformers import RobertaTokenizerFast
    tok_file = os.path.join(str(name_or_dir), "tokenizer.json")
    if os.path.isfile(tok_file):
        tok = RobertaTokenizerFast(tokenizer_file=tok_file)
    else:
        tok = RobertaTokenizerFast.from_pretrained(name_or_dir, local_files_only=True)
    pre = json.loads(tok.backend_tokenizer.to_str()).get("pre_tokenizer") or {}
    if pre.get("type") != "ByteLevel":
        raise RuntimeError(f"tokenizer from {name_or_dir!r} has pre-tokenizer {pre.get('type')!r}, RoBERTa's is ByteLevel")
    return tok


def encode_expressions(tokenizer, text_queries):
    """The reference's `tokenizer.batch_encode_plus(text_queries, padding='longest', return_tensors='pt')`
    (models/soc.py:104-106,168-169) -> (input_ids, attention_mask) int64 [B,L].  Spelled as the tokenizer's `__call__`,
    which is the same method in transformers 4.x and the only one left in 5.x (`batch_encode_plus` is gone there)."""
    tok = tokenizer(list(text_queries), padding="longest", return_tensors="pt")
    return tok["input_ids"], tok["attention_mask"]


class SOC(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.backbone not in ("video-swin-t", "video-swin-s", "video-swin-b"):
            raise NotImplementedError("only the Video-Swin backbones are on the MI355X hot path "
                                      "(resnet50 branch of the reference: SURVEY.md section 2, out of scope)")
        self.backbone = build_video_swin_backbone(config)
        dt = config.DeformTransformer
        self.num_feature_levels = dt["num_feature_levels"]
        d_model = dt["d_model"]
        self.num_queries = dt["num_queries"]
        self.rel_coord = config.rel_coord
        if not self.rel_coord:
            raise NotImplementedError("K4 is built for rel_coord=True (every shipped config)")
        self.transformer = build_deforamble_transformer(dt)

        chans = self.backbone.num_channels[-3:]
        proj = [nn.Sequential(nn.Conv2d(c, d_model, kernel_size=1), nn.GroupNorm(32, d_model)) for c in chans]
        c_in = chans[-1]
        for _ in range(self.num_feature_levels - len(chans)):
            proj.append(nn.Sequential(nn.Conv2d(c_in, d_model, kernel_size=3, stride=2, padding=1),
                                      nn.GroupNorm(32, d_model)))
            c_in = d_model
        self.input_proj = nn.ModuleList(proj)
        for p in self.input_proj:
            nn.init.xavier_uniform_(p[0].weight, gain=1)
            nn.init.zeros_(p[0].bias)

        n_pred = self.transformer.decoder.num_layers
        box, cls = MLP(d_model, d_model, 4, 3), nn.Linear(d_model, config.num_classes)
        cls.bias.data.fill_(-math.log((1 - 0.01) / 0.01))
        nn.init.zeros_(box.layers[-1].weight)
        nn.init.zeros_(box.layers[-1].bias)
        if not config.with_box_refine:
            raise NotImplementedError("with_box_refine=False is not used by any shipped config")
        self.class_embed = nn.ModuleList(copy.deepcopy(cls) for _ in range(n_pred))
        self.bbox_embed = nn.ModuleList(copy.deepcopy(box) for _ in range(n_pred))
        self.bbox_embed[0].layers[-1].bias.data[2:].fill_(-2.0)
        self.transformer.decoder.bbox_embed = self.bbox_embed  # shared modules, aliased state_dict keys

        self.text_encoder, self.tokenizer = _load_text_stack(config)
        accelerate_text_encoder(self.text_encoder)     # GPU form of the encoder layers (text_fast.py); no-op on CPU
        self.freeze_text_encoder = config.freeze_text_encoder
        self.text_pos = PositionEmbeddingSine1D(d_model, normalize=True)
        self.query_embed = nn.Embedding(self.num_queries, d_model)
        self.spatial_decoder = FPNSpatialDecoder(d_model, 2 * [d_model] + [self.backbone.num_channels[0]],
                                                 config.mask_kernels_dim)
        self.voc = VOC(config.VOC)
        self.vlf = MMF(d_model=d_model, nhead=8)
        self.lvf = MMF(d_model=d_model, nhead=8)
        self.txt_proj = FeatureResizer(self.text_encoder.config.hidden_size, d_model, dropout=0.1)

        self.controller_layers = config.controller_layers
        self.in_channels = config.mask_kernels_dim
        self.dynamic_mask_channels = config.dynamic_mask_channels
        self.mask_out_stride = self.mask_feat_stride = 4
        c, ch, nl = self.in_channels, self.dynamic_mask_channels, self.controller_layers
        self.weight_nums = [(c + 2) * ch] + [ch * ch] * (nl - 2) + [ch]
        self.bias_nums = [ch] * (nl - 1) + [1]
        self.num_gen_params = sum(self.weight_nums) + sum(self.bias_nums)
        if (c, ch, nl) != (8, 8, 3):
            raise NotImplementedError("K4 is built for mask_kernels_dim=8, dynamic_mask_channels=8, "
                                      "controller_layers=3 (every shipped config)")
        self.controller = MLP(d_model, d_model, self.num_gen_params, 3)
        for layer in self.controller.layers:
            nn.init.zeros_(layer.bias)
            nn.init.xavier_uniform_(layer.weight)
        self.vl_loss, self.aux_loss = config.vl_loss, config.aux_loss
        # "split" | "f32" | None (= hot_ops.DEFAULT_MATMUL_MODE, i.e. SOC_MATMUL or "split"): the arithmetic of THIS model's
        # large products; an attribute of the model, handed to every launch as an argument (ABI 16), not process state
        self.matmul_mode = getattr(config, "matmul_mode", None)

    def _side_stream(self, device):
        """One side stream per (device, calling stream): forwards issued on different streams (several
        clips in flight) fork onto different side streams and stay independent."""
        streams = self.__dict__.setdefault("_streams", {})
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        if key not in streams:
            # default priority on purpose: a high-priority text / coarse-level branch delays the chip-filling kernels beside
            # it (measured 6.28 -> 6.83 ms per clip; only the pipelined TAIL branch gains from priority, graph_runner.py)
            streams[key] = torch.cuda.Stream(device=device)
        return streams[key]

    # ------------------------------------------------------------------ text
    def forward_text(self, text_queries, device):
        """list[str] (needs tokenizer files) or pre-tokenised {'input_ids','attention_mask'} [B,L]."""
        if isinstance(text_queries, (list, tuple)) and text_queries and isinstance(text_queries[0], str):
            if self.tokenizer is None:
                raise RuntimeError("no tokenizer files available offline: pass pre-tokenised "
                                   "{'input_ids', 'attention_mask'} tensors instead of strings")
            ids, attn = encode_expressions(self.tokenizer, text_queries)
        else:
            ids, attn = text_queries["input_ids"], text_queries["attention_mask"]
        ids, attn = ids.to(device), attn.to(device)
        enc = self.text_encoder(input_ids=ids, attention_mask=attn)
        words = self.txt_proj(enc.last_hidden_state.transpose(0, 1))   # [L,B,C]
        sentence = self.txt_proj(enc.pooler_output)                     # [B,C]
        return NestedTensor(words, attn.ne(1)), sentence

    # ------------------------------------------------------------------ forward
    def _project_level(self, l: int, src, B: int, T: int, clip_major: bool = False):
        """input_proj[l] = Conv2d(1x1) + GroupNorm(32) of a backbone level, returned as the '(t h w) b c'
        sequence the fusion consumes (reference models/soc.py:226-230).  The backbone's maps are
        channels-last in memory, so on the GPU the 1x1 convolution is a GEMM over tokens and the
        GroupNorm runs on the token-major result (K10): no layout copy before or after.
        clip_major: return 'b (t h w) c' instead -- a view for any B (the fusion then runs batch-first)."""
        conv, gn = self.input_proj[l][0], self.input_proj[l][1]
        n, cin, h, w = src.shape
        tok = src.permute(0, 2, 3, 1)                              # a view of the backbone's native layout
        if not (src.is_cuda and tok.is_contiguous() and conv.kernel_size == (1, 1)):
            return (self._seq_clip_major if clip_major else self._seq)(self.input_proj[l](src), B, T)
        y = fused.linear(tok.reshape(n, h * w, cin), conv.weight.view(conv.out_channels, cin), conv.bias)   # K13b / K20 / library
        y = hot_ops.groupnorm_tokens(y, gn.weight, gn.bias, gn.num_groups, gn.eps)    # '(b t) (h w) c'
        c = y.shape[-1]
        if clip_major:
            return y.view(B, T * h * w, c)
        return y.view(B, T, h * w, c).permute(1, 2, 0, 3).reshape(T * h * w, B, c)  # a view for B = 1

    @staticmethod
    def _seq(x, B, T):
        """'(b t) c h w -> (t h w) b c'"""
        _, c, h, w = x.shape
        return x.view(B, T, c, h, w).permute(1, 3, 4, 0, 2).reshape(T * h * w, B, c)

    @staticmethod
    def _seq_clip_major(x, B, T):
        """'(b t) c h w -> b (t h w) c'"""
        _, c, h, w = x.shape
        return x.permute(0, 2, 3, 1).reshape(B, T * h * w, c)

    @staticmethod
    def _tokens(x, B, T, h, w):
        """'(t h w) b c -> (b t) (h w) c' -- a free view for B = 1"""
        c = x.shape[-1]
        return x.view(T, h * w, B, c).permute(2, 0, 1, 3).reshape(B * T, h * w, c)

    @torch.no_grad()
    def forward(self, samples: NestedTensor, valid_indices, text_queries, targets):
        """samples.tensors [T,B,3,H,W] + mask [T,B,H,W]; text_queries: B strings (or pre-tokenised);
        targets[0][b]['size'] = (H,W) of the model input.  Returns the reference's output dict:
        pred_masks [T,B,Q,H/4,W/4], pred_cls [T,B,Q,K], pred_boxes [T,B,Q,4], pred_logit [B,Q,C],
        text_sentence_feature [B,C], aux_outputs []."""
        return self.forward_tail(self.forward_head(samples, valid_indices, text_queries), targets)

    @torch.no_grad()
    def forward_head(self, samples: NestedTensor, valid_indices, text_queries):
        """First part of forward: text encoder ‖ Video-Swin, vision-language fusion, deformable encoder.  Returns the
        state forward_tail needs; graph_runner.PipelinedClipGraph runs the tail of clip i beside the head of clip i+1."""
        return self.forward_fuse_encode(self.forward_backbone(samples, valid_indices, text_queries))

    @torch.no_grad()
    @_in_model_mode
    def forward_backbone(self, samples: NestedTensor, valid_indices, text_queries, fork: bool = True):
        """Stage A: RoBERTa ‖ Video-Swin.  -> state for forward_fuse_encode."""
        if self.training:
            raise RuntimeError("this build of SOC is inference-only: call model.eval()")
        if valid_indices is not None:
            raise NotImplementedError("valid_indices is only used by the A2D/JHMDB loaders "
                                      "(reference soc.py:208-215), outside the inference hot path")
        device = samples.tensors.device
        # The text branch (RoBERTa: ~150 latency-bound launches on 10 tokens) and the video backbone
        # are independent until the first vision-language fusion: run them on two HIP streams so the
        # small text kernels hide under the Swin kernels (also captured as a fork/join in ClipGraph).
        side = self._side_stream(device) if (device.type == "cuda" and fork) else None
        if side is not None:
            main = torch.cuda.current_stream(device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                tx = self.forward_text_state(text_queries, device)
            vs = self.forward_video(samples)
            main.wait_stream(side)
        else:
            tx = self.forward_text_state(text_queries, device)
            vs = self.forward_video(samples)
        return {**vs, **tx}

    @torch.no_grad()
    @_in_model_mode
    def forward_text_state(self, text_queries, device):
        """The text half of stage A: RoBERTa + FeatureResizer + the words' position encoding (reference models/soc.py:167-181
        and the `text_pos` of :232-233).  graph_runner.TwoStreamClipGraph runs it on its auxiliary stream, in front of the tail
        of the previous clip, instead of beside Video-Swin."""
        text, sentence = self.forward_text(text_queries, device)
        text_pos = self.text_pos(text).permute(2, 0, 1)     # ten tiny launches: on the text branch, not in front of the fusion
        words, word_pad = text.decompose()
        return {"words": words, "word_pad": word_pad, "text_pos": text_pos, "sentence": sentence}

    @torch.no_grad()
    @_in_model_mode
    def forward_video(self, samples: NestedTensor):
        """The video half of stage A: Video-Swin + position encodings (rewrites samples to '(b t)' like the reference)."""
        backbone_out, pos = self.backbone(samples)
        return {"feats": [f.tensors for f in backbone_out], "masks": [f.mask for f in backbone_out], "pos": pos,
                "sample_mask": samples.mask, "unpadded": bool(getattr(samples, "unpadded", False))}

    @torch.no_grad()
    @_in_model_mode
    def forward_fuse_encode(self, sa, fork: bool = True):
        """Stage B: input_proj + vision-language fusion of every level, deformable encoder.  -> state for forward_tail."""
        feats, fmasks, pos = sa["feats"], sa["masks"], sa["pos"]
        words, word_pad, sentence, unpadded = sa["words"], sa["word_pad"], sa["sentence"], sa["unpadded"]
        device = words.device
        side = self._side_stream(device) if (device.type == "cuda" and fork) else None
        main = torch.cuda.current_stream(device) if side is not None else None
        B = words.shape[1]
        T = pos[-1].shape[0] // B
        text_pos = sa["text_pos"]
        # A launch group (B > 1): '(t h w) b c' would be a real permute of every level (118 MB each way at level 0 for four
        # clips); the fusion is a cross-attention per clip, so it runs batch-first on the 'b (t h w) c' view of the tokens.
        clip_major = B > 1 and device.type == "cuda" and os.environ.get("SOC_GROUP_SEQ_FIRST") != "1"
        if clip_major:
            words, text_pos = words.transpose(0, 1).contiguous(), text_pos.transpose(0, 1).contiguous()
        to_seq = self._seq_clip_major if clip_major else self._seq

        levels = list(zip(feats[-3:], fmasks[-3:], pos[-3:]))
        n_levels = self.num_feature_levels
        if n_levels > len(levels) + 1:
            raise NotImplementedError("more than one extra feature level is not used by any shipped config")

        def fuse_level(l):
            """input_proj + vision<-language fusion of level l -> (tokens '(b t) (h w) c', mask, pos, lang | None)"""
            lang = None
            if l < len(levels):
                src, mask, pos_l = levels[l]
                h, w = src.shape[-2:]
                seq = self._project_level(l, src, B, T, clip_major)
                if l == len(levels) - 1:  # only langs[-1] is read downstream
                    lang = self.lvf(tgt=words, memory=seq,
                                    memory_key_padding_mask=mask.view(B, T, h, w).reshape(B, -1),
                                    pos=to_seq(pos_l, B, T), batch_first=clip_major)
                    if clip_major:
                        lang = lang.transpose(0, 1).contiguous()         # [L,B,C] like the sequence-first path
            else:                          # the extra level: 3x3 / stride-2 conv of the coarsest backbone map
                src = self.input_proj[l](feats[-1])
                mask = resize_pad_mask(sa["sample_mask"], src.shape[-2:], unpadded)
                pos_l = self.backbone.position_encoding(NestedTensor(src, mask), unpadded)
                h, w = src.shape[-2:]
                seq = to_seq(src, B, T)
            fused = self.vlf(tgt=seq, memory=words, memory_key_padding_mask=word_pad, pos=text_pos, batch_first=clip_major)
            if clip_major:
                return fused.reshape(B * T, h * w, fused.shape[-1]), mask, pos_l, lang
            return self._tokens(fused, B, T, h, w), mask, pos_l, lang

        # The levels are independent of each other.  The two finest ones (94 % of the tokens: chip-filling GEMMs) run on
        # the main stream, the coarse ones -- ~50 short, latency-bound launches, longer end to end than level 0 alone --
        # beside them on the side stream (rocprofv3 trace at the BASELINE config: 0.30 ms against 0.32 ms; with levels
        # 1-3 on the side stream it was 0.20 ms against 0.46 ms).
        per_level = [None] * n_levels
        if side is not None and n_levels > 2:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for l in range(n_levels - 1, 1, -1):
                    per_level[l] = fuse_level(l)
            for l in (0, 1):
                per_level[l] = fuse_level(l)
            main.wait_stream(side)
        else:
            per_level = [fuse_level(l) for l in range(n_levels)]
        srcs = [p[0] for p in per_level]
        masks = [p[1] for p in per_level]
        poses = [p[2] for p in per_level]
        lang_last = next(p[3] for p in per_level if p[3] is not None)
        _, ctx = self.transformer.encode(srcs, masks, poses, token_major=True, unpadded=unpadded, maps=False)
        return {"ctx": ctx, "feats0": feats[0], "lang_last": lang_last, "word_pad": word_pad,
                "sentence": sentence, "B": B, "T": T}

    @torch.no_grad()
    @_in_model_mode
    def forward_tail(self, state, targets, fork: bool = True, voc_per_clip: bool = False):
        """Second part of forward: FPN spatial decoder ‖ query decoder -> VOC -> heads, dynamic mask head.
        ``fork=False`` keeps everything on the calling stream (used when the caller already runs the tail beside
        another clip's head).
        ``voc_per_clip=True``: the VOC module clusters every clip over its own frames (VOC.forward(independent_clips=True): one
        batched pass that computes what B calls with B = 1 compute).  VOC is the ONE place where the reference's forward couples
        the clips of a batch -- it reshapes [T,B,Q,C] to [B,T,Q,C] instead of permuting (models/voc.py:279), which deals the
        frames of a batch over its clips; measured: with VOC per clip a B = 2 forward gives each clip its B = 1 outputs to 5e-5,
        without it they differ by 4.9 on a logit scale of 37 (tools/experiments/batch2_voc_probe.py) -- so a caller that batches
        INDEPENDENT clips and wants the reference's per-clip (B = 1) results (the group pipelines of graph_runner) sets it; the
        default reproduces the reference's batch semantics (golden padded_b2_forward.npz)."""
        ctx, feats0, lang_last = state["ctx"], state["feats0"], state["lang_last"]
        word_pad, sentence, B, T = state["word_pad"], state["sentence"], state["B"], state["T"]
        device = feats0.device
        Q = self.num_queries
        tgt = lang_last.new_zeros(B, T, Q, lang_last.shape[-1])
        use_tokens = self.spatial_decoder.tokens_supported(ctx[0])     # GPU: the FPN runs token-major, no NCHW maps

        def run_fpn():
            if use_tokens:
                return self.spatial_decoder.forward_tokens(ctx[0], ctx[6][:self.num_feature_levels - 1], feats0)
            memory = self.transformer.memory_maps(ctx)
            return self.spatial_decoder(memory[-1], [memory[1], memory[0], feats0])  # '(b t) 8 h/4 w/4'

        # The FPN spatial decoder (convs over the memory maps) only meets the query branch (decoder -> VOC ->
        # heads -> controller, ~150 small latency-bound launches) at the dynamic mask head, so it runs on the
        # side stream meanwhile.
        side = self._side_stream(device) if (device.type == "cuda" and fork) else None
        if side is not None:
            main = torch.cuda.current_stream(device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fpn = run_fpn()
        else:
            fpn = run_fpn()

        hs, init_ref, inter_refs = self.transformer.decode(ctx, tgt, self.query_embed.weight)

        # text feature the reference reports for its vl-loss: mean over real words of lvf's output
        keep = (~word_pad).to(lang_last.dtype).transpose(0, 1)[..., None]   # [L,B,1]
        text_feature = (lang_last * keep).sum(0) / keep.sum(0)

        C = hs.shape[-1]
        hs_t = hs.view(hs.shape[0], B, T, Q, C).transpose(1, 2)              # l t b q c
        # voc_per_clip: every clip clustered over its own frames -- one batched pass, bit-compatible with B separate B = 1 calls
        # (the reference reshapes [T,B,Q,C] to [B,T,Q,C] instead of permuting, which deals the frames of a batch over its
        # clips: models/voc.py:279; identity at B = 1)
        voc_hs = self.voc(hs_t, sentence, independent_clips=bool(voc_per_clip))     # [1,B,Q,C]
        hs0 = hs[0].view(B, T, Q, C) + voc_hs[0][:, None]                    # level 0 (b t q c)

        cls = self.class_embed[0](hs0)
        box = self.bbox_embed[0](hs0)
        box = hot_ops.box_refine(box.view(B * T, Q, 4), init_ref)[0].view(B, T, Q, 4)
        params = self.controller(hs0)                                        # b t q 169
        refs = inter_refs[0][..., :2].reshape(B, T * Q, 2)
        if side is not None:
            main.wait_stream(side)
        hm, wm = fpn.shape[-2:]
        fpn = fpn.view(B, T, fpn.shape[1], hm, wm)
        per_b = []
        for b in range(B):
            size = targets[0][b]["size"]
            img_h, img_w = (float(v) for v in (size.tolist() if torch.is_tensor(size) else size))
            m = hot_ops.dynamic_mask(fpn[b], params[b].reshape(T * Q, -1), refs[b], (img_h, img_w),
                                     self.mask_feat_stride)
            per_b.append(m.view(T, Q, hm, wm))
        pred_masks = torch.stack(per_b, 1)                                   # t b q h w

        out = {"pred_masks": pred_masks,
               "pred_logit": voc_hs[0],
               "pred_boxes": box.transpose(0, 1),
               "text_sentence_feature": text_feature,
               "pred_cls": cls.transpose(0, 1)}
        if self.aux_loss:
            out["aux_outputs"] = []
        return out

    @staticmethod
    def split_state(state):
        """The hand-over state of a B-clip head as B single-clip states (views, no copies): what forward_tail needs to run
        per clip.  The head (Video-Swin, fusion, deformable encoder) treats the clips of a batch independently -- every op is
        per token, per window or per frame -- so a head over two clips fills the chip better at the small stages and gives
        each clip what a B = 1 head gives it (to f32 rounding).  The REFERENCE's tail does not: its B = 2 outputs of a clip
        differ from its B = 1 outputs by 5 % of the logit scale (measured on the reference itself), so a
        clip-parallel caller that wants the reference's per-clip (B = 1) results batches the head only."""
        B = state["B"]
        if B == 1:
            return [state]
        return [SOC.slice_state(state, b, b + 1) for b in range(B)]

    @staticmethod
    def slice_state(state, b0: int, b1: int):
        """Clips b0 .. b1 - 1 of a batched head's hand-over state (views, no copies): a sub-group forward_tail can run on."""
        T = state["T"]
        memory, spatial_shapes, level_start, ratios, mask, pad_flag, shapes = state["ctx"]
        rows = slice(b0 * T, b1 * T)
        ctx = (memory[rows], spatial_shapes, level_start, ratios[rows], mask[rows], pad_flag, shapes)
        return {"ctx": ctx, "feats0": state["feats0"][rows], "lang_last": state["lang_last"][:, b0:b1],
                "word_pad": state["word_pad"][b0:b1], "sentence": state["sentence"][b0:b1], "B": b1 - b0, "T": T}

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


def build(args):
    """-> (model, criterion, postprocessor) like reference models/soc.py:624-646.  The criterion is
    training-only (out of scope) and inference callers discard it (infer_refytb.py:133): None."""
    model = SOC(args)
    from .gemm_tuning import enable_tuned_gemms
    enable_tuned_gemms()  # no-op without a GPU / table
    return model, None, build_postprocessors(args.dataset_name)


def build_model(args):
    return build(args)
