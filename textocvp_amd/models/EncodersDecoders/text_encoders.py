"""
Transformer text encoder of TextOCVP_CustomTF.
Reference: models/EncodersDecoders/text_encoders.py:14-138 (forward :89-125).
``CustomTokenizer`` (:142-194) is data-side host logic; a small counterpart is kept here only so that the
reference's ``data/*.py`` imports resolve through the ``dropin/models`` alias package.
"""

import re

import torch
import torch.nn as nn

from ... import kernels as K

__all__ = ["TransformerTextEncoder", "CustomTokenizer"]


class TransformerTextEncoder(nn.Module):
    """
    token + position embedding -> LayerNorm(eps 1e-8) -> zero padding rows -> 2x post-norm
    transformer encoder layers (GELU, key-padding mask) -> LayerNorm + Linear to the token dim.

    ``nn.TransformerEncoder`` is instantiated ONLY as the parameter container that yields the
    reference's state_dict keys (transformer.layers.N.self_attn.in_proj_weight, ...); it is never
    called.  The arithmetic runs on the HIP kernels: fused embedding/LayerNorm front end, fp32-MFMA
    GEMMs with bias/GELU/residual epilogues, attention with per-sample key lengths.
    Padded positions keep finite, non-zero outputs and are attended to by the predictor's
    cross-attention, exactly as in the reference.
    """

    def __init__(self, input_dim, num_layers, num_heads, output_dim, vocab_size,
                 context_length=50, dropout=0.1):
        super().__init__()
        self.vocab_size = vocab_size
        self.padding_idx = 0
        self.num_heads = num_heads
        layer = nn.TransformerEncoderLayer(d_model=input_dim, nhead=num_heads,
                                           dim_feedforward=input_dim * 4, dropout=dropout,
                                           activation="gelu")
        self.transformer = nn.TransformerEncoder(layer, num_layers, enable_nested_tensor=False)
        self.token_embedding = nn.Embedding(vocab_size, input_dim)
        self.position_embedding = nn.Embedding(context_length, input_dim)
        self.layer_norm = nn.LayerNorm(input_dim, eps=1e-8, elementwise_affine=True)
        self.dropout = nn.Dropout(p=dropout)          # inactive: inference only
        self.text_out_projection = nn.Sequential(
            nn.LayerNorm(input_dim), nn.Linear(input_dim, output_dim))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(module):
        """ N(0, 0.02) for linear / attention / embedding weights (text_encoders.py:73-87) """
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=0.02)
        elif isinstance(module, nn.MultiheadAttention):
            module.in_proj_weight.data.normal_(mean=0.0, std=0.02)
            module.out_proj.weight.data.normal_(mean=0.0, std=0.02)

    def forward(self, text, text_length):
        """ text (B, L) int64 token ids, text_length (B,) -> (B, L, output_dim) """
        dev = self.token_embedding.weight.device
        text = text.to(dev).contiguous()
        B, L = text.shape
        if L > self.position_embedding.num_embeddings:
            raise ValueError(f"caption length {L} exceeds the position table")
        key_len = text_length.to(device=dev, dtype=torch.int32).contiguous()
        x = K.text_embed(text, self.token_embedding.weight, self.position_embedding.weight,
                         self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        E = x.shape[-1]
        for layer in self.transformer.layers:
            sa = layer.self_attn
            qkv = K.linear(x, sa.in_proj_weight, sa.in_proj_bias)
            a = K.mha(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], self.num_heads,
                      (E // self.num_heads) ** -0.5, key_len=key_len)
            x = K.layer_norm(K.linear(a, sa.out_proj.weight, sa.out_proj.bias, residual=x),
                             layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
            h = K.linear(x, layer.linear1.weight, layer.linear1.bias, act=K.ACT_GELU)
            x = K.layer_norm(K.linear(h, layer.linear2.weight, layer.linear2.bias, residual=x),
                             layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
        ln, proj = self.text_out_projection[0], self.text_out_projection[1]
        return K.linear(K.layer_norm(x, ln.weight, ln.bias, ln.eps), proj.weight, proj.bias)


class CustomTokenizer:
    """
    Word-level caption tokenizer with the reference's surface (text_encoders.py:142-194): ``[CLS]`` + word
    ids + ``[SEP]``, batches right-padded with ``[PAD]``.  Host-side integer bookkeeping (no kernel).  Words are
    split with ``nltk.word_tokenize`` when nltk is installed (as in the reference), else on word / punctuation
    boundaries -- identical on the CATER / CLIPort caption grammars (lower-case words, commas, full stops).
    """

    def __init__(self, vocabulary):
        assert "[PAD]" in vocabulary, "Vocabulary must contain '[PAD]' token..."
        self.padding_idx = vocabulary["[PAD]"]
        self.vocabulary = vocabulary
        self.vocabulary_reverse = {v: k for k, v in vocabulary.items()}
        self._split = self._resolve_backend()

    _warned = False

    @classmethod
    def _resolve_backend(cls):
        """
        The word splitter, resolved ONCE: nltk.word_tokenize when nltk and its punkt data are usable (the
        reference's splitter), else a regular expression -- with ONE warning, because the two differ on
        contractions, hyphens, quotes and numbers (identical only on the CATER / CLIPort caption grammars).
        Only the two "not installed" failures select the fallback; any other nltk error surfaces.
        """
        try:
            import nltk
            nltk.word_tokenize("probe , sentence .")
            return nltk.word_tokenize
        except (ImportError, LookupError) as err:           # nltk absent / its punkt data absent
            if not cls._warned:
                import warnings
                warnings.warn(f"CustomTokenizer: nltk.word_tokenize unavailable ({type(err).__name__}: {err}); "
                              "splitting captions on word / punctuation boundaries instead -- identical on the CATER / "
                              "CLIPort grammars, different on contractions, hyphens, quotes and numbers")
                cls._warned = True
            return lambda text: re.findall(r"\w+|[^\w\s]", text)

    def _words(self, text):
        return self._split(text)

    def text2tokens(self, x):
        return [self.vocabulary[w] for w in self._words(x)]

    def tokenize(self, caption):
        ids = [self.vocabulary["[CLS]"]] + self.text2tokens(caption) + [self.vocabulary["[SEP]"]]
        return torch.tensor(ids, dtype=torch.long), torch.tensor(len(ids), dtype=torch.long)

    def tokenize_batch(self, caption):
        toks, lens = zip(*(self.tokenize(c) for c in caption))
        toks = torch.nn.utils.rnn.pad_sequence(list(toks), batch_first=True, padding_value=self.padding_idx)
        return toks, torch.stack(lens)

    def tokens2text(self, tokens):
        return "".join(" " + self.vocabulary_reverse[int(t)] for t in tokens)
