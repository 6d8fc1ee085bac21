"""Text side of the model: RoBERTa encoder wrapper + the two feature resizers.

Reference: models/text_encoder/text_encoder.py (FeatureResizer :16-29, TextEncoder :32-83).  The reference loads
HF `RobertaModel` / `RobertaTokenizerFast` from a local `checkpoints/roberta-base` directory that is not part of the
repository.  Here:
  * if that directory exists it is used exactly like the reference does;
  * otherwise the encoder is a random-initialised RoBERTa-base (HF `RobertaConfig`, same state_dict names) and
    captions are tokenised by a deterministic whitespace/hash tokenizer (synthetic benchmark use only);
  * callers may also pass PRECOMPUTED text features (a `PrecomputedText`), which is what the synthetic
    BASELINE configs #1-#4 use ("random text tokens").
HF transformers is the same third-party dependency the reference uses; its arithmetic is not restated here.
"""
import os
import zlib
from typing import NamedTuple

import torch
from torch import nn

from .. import amp_cache


class PrecomputedText(NamedTuple):
    """Stand-in for the encoder output: features [B,L,768], sentence [B,768], pad_mask [B,L] (True = padding)."""
    features: torch.Tensor
    sentence: torch.Tensor
    pad_mask: torch.Tensor


class FeatureResizer(nn.Module):
    """Linear -> LayerNorm(eps 1e-12) -> dropout."""

    def __init__(self, input_feat_size, output_feat_size, dropout, do_ln=True):
        super().__init__()
        self.do_ln = do_ln
        self.fc = amp_cache.Linear(input_feat_size, output_feat_size, bias=True)
        self.layer_norm = nn.LayerNorm(output_feat_size, eps=1e-12)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        x = self.fc(x)
        if self.do_ln:
            x = self.layer_norm(x)
        return self.dropout(x)


class HashTokenizer(nn.Module):
    """Deterministic fallback tokenizer (no vocabulary files): <s>=0, </s>=2, <pad>=1, words hashed into [3, vocab)."""

    def __init__(self, vocab_size=50265):
        super().__init__()
        self.vocab_size = vocab_size

    def forward(self, texts):
        rows = [[0] + [3 + zlib.crc32(w.encode()) % (self.vocab_size - 3) for w in t.lower().split()] + [2] for t in texts]
        n = max(len(r) for r in rows)
        ids = torch.full((len(rows), n), 1, dtype=torch.long)
        att = torch.zeros((len(rows), n), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r)
            att[i, : len(r)] = 1
        return {"input_ids": ids, "attention_mask": att}


class _HFTokenizer(nn.Module):
    def __init__(self, path):
        super().__init__()
        from transformers import RobertaTokenizerFast
        self.tokenizer = RobertaTokenizerFast.from_pretrained(path)

    def forward(self, texts):
        return dict(self.tokenizer(list(texts), padding="longest", return_tensors="pt"))


class TextEncoder(nn.Module):
    CKPT_DIR = "checkpoints/roberta-base"

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.hidden_dim = args.hidden_dim
        self.text_backbone_name = args.text_backbone
        assert self.text_backbone_name == "Roberta", f'error: Text Encoder "{self.text_backbone_name}" is not supported'
        self.feat_dim = 768
        self.freeze_text_encoder = args.freeze_text_encoder
        self.tokenizer = None
        self.text_backbone = None
        if getattr(args, "text_encoder_lazy", False):
            return      # synthetic / fixture use: features are supplied by the caller, no RoBERTa is built
        from transformers import RobertaConfig, RobertaModel
        if os.path.isdir(self.CKPT_DIR):
            self.tokenizer = _HFTokenizer(self.CKPT_DIR)
            self.text_backbone = RobertaModel.from_pretrained(self.CKPT_DIR)
        else:
            cfg = RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                                pad_token_id=1, bos_token_id=0, eos_token_id=2)
            self.tokenizer = HashTokenizer(cfg.vocab_size)
            self.text_backbone = RobertaModel(cfg)
        if self.freeze_text_encoder:
            for p in self.text_backbone.parameters():
                p.requires_grad_(False)

    def _encode(self, texts, device):
        tok = {k: v.to(device) for k, v in self.tokenizer(texts).items()}
        enc = self.text_backbone(**tok)
        return enc.last_hidden_state, enc.pooler_output, tok["attention_mask"].ne(1).bool()

    def forward(self, texts, device):
        if isinstance(texts, PrecomputedText):
            return texts.features.to(device), texts.sentence.to(device), texts.pad_mask.to(device)
        if self.text_backbone is None:
            raise RuntimeError("TextEncoder was built lazily (no RoBERTa): pass a PrecomputedText")
        if self.freeze_text_encoder:
            with torch.no_grad():
                return self._encode(texts, device)
        return self._encode(texts, device)

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
