"""Synthetic stand-ins for what is not available offline: a Ref-YouTube-VOS / Ref-DAVIS shaped dataset
directory (JPEG frames + meta_expressions.json) and a vocabulary-free tokenizer.  For smoke runs,
tests and pipeline timing only -- never a substitute for the real tokenizer or data in evaluation."""
from __future__ import annotations

import json
import os
from typing import Sequence

import numpy as np
import torch

WORDS = ("a", "the", "person", "dog", "cat", "car", "left", "right", "walking", "running", "red", "white", "black",
         "small", "large", "in", "on", "front", "behind", "of", "with", "holding", "riding", "standing", "jumping")


class HashTokenizer:
    """str -> int64 [1, L] = <s> + one id per whitespace word (FNV-1a into [3, vocab)) + </s>.
    Deterministic and vocabulary-free; RoBERTa's real BPE needs files that cannot be fetched here."""

    def __init__(self, vocab_size: int = 50265):
        self.vocab_size = vocab_size

    def __call__(self, text: str) -> torch.Tensor:
        ids = [0]
        for word in text.split():
            h = 0xCBF29CE484222325
            for ch in word.encode():
                h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
            ids.append(3 + h % (self.vocab_size - 3))
        ids.append(2)
        return torch.tensor([ids], dtype=torch.long)


def write_synthetic_roberta_tokenizer(path: str, words: Sequence[str] = WORDS) -> str:
    """A directory `RobertaTokenizerFast.from_pretrained` loads offline: byte-level BPE with RoBERTa's special tokens
    (<s> 0, <pad> 1, </s> 2, <unk> 3), RoBERTa's `<s> ... </s>` post-processing, and merges that build every word of
    `words` (with its leading-space form) left to right.  The real roberta-base vocabulary cannot be fetched here; this one
    exercises the same code path (models/soc.py:104-106,167-169) with ids of a different vocabulary."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for ch in sorted(pre_tokenizers.ByteLevel.alphabet()):
        vocab[ch] = len(vocab)
    merges = []

    def merge(a, b):
        if (a, b) not in merges:
            merges.append((a, b))
            vocab.setdefault(a + b, len(vocab))
    for word in words:
        left = word[0]
        for ch in word[1:]:
            merge(left, ch)
            left += ch
    for word in words:                                  # "\u0120" is the byte-level image of the space in front of a word
        merge("\u0120", word)
    vocab["<mask>"] = len(vocab)
    tk = Tokenizer(models.BPE(vocab=vocab, merges=merges, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    tk.post_processor = processors.RobertaProcessing(sep=("</s>", 2), cls=("<s>", 0), trim_offsets=True, add_prefix_space=False)
    os.makedirs(path, exist_ok=True)
    tk.save(os.path.join(path, "tokenizer.json"))
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(path, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n" + "\n".join(a + " " + b for a, b in merges) + "\n")
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "RobertaTokenizerFast", "bos_token": "<s>", "eos_token": "</s>", "pad_token": "<pad>",
                   "unk_token": "<unk>", "cls_token": "<s>", "sep_token": "</s>", "mask_token": "<mask>",
                   "model_max_length": 512}, f)
    return path


def _frame(rng, h, w, t):
    """blocky moving pattern + noise: compresses like a natural JPEG, differs per frame"""
    cells = rng.integers(0, 256, (h // 16 + 5, w // 16 + 10, 3))      # margin for t <= 64
    img = cells.repeat(16, 0).repeat(16, 1)[t:t + h, 2 * t:2 * t + w]
    return np.clip(img + rng.integers(-12, 13, (h, w, 3)), 0, 255).astype(np.uint8)


def make_dataset(root: str, videos: int = 2, frames: int = 8, height: int = 720, width: int = 1280,
                 expressions: int = 2, seed: int = 0, split: str = "valid", quality: int = 90,
                 words: Sequence[str] = WORDS, n_words: int = 0) -> str:
    """Writes <root>/<split>/JPEGImages/<video>/<%05d>.jpg and <root>/meta_expressions/<split>/meta_expressions.json.
    n_words > 0 fixes the expression length (one token count -> one hipGraph geometry).  `expressions`: a count, or one count
    per video (the real sets are ragged: infer_refytb.py:185 loops over however many expressions a video has)."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    meta = {"videos": {}}
    for v in range(videos):
        name = f"video{v:03d}"
        folder = os.path.join(root, split, "JPEGImages", name)
        os.makedirs(folder, exist_ok=True)
        names = [f"{5 * t:05d}" for t in range(frames)]
        vr = np.random.default_rng(seed * 1000 + v)
        for t, n in enumerate(names):
            Image.fromarray(_frame(np.random.default_rng(seed * 1000 + v), height, width, t)).save(
                os.path.join(folder, n + ".jpg"), quality=quality)
        exps = {}
        for e in range(expressions if isinstance(expressions, int) else expressions[v]):
            k = n_words or int(vr.integers(3, 9))
            exps[str(e)] = {"exp": " ".join(words[int(i)] for i in rng.integers(0, len(words), k))}
        meta["videos"][name] = {"frames": names, "expressions": exps}
    mdir = os.path.join(root, "meta_expressions", split)
    os.makedirs(mdir, exist_ok=True)
    with open(os.path.join(mdir, "meta_expressions.json"), "w") as f:
        json.dump(meta, f)
    return root
