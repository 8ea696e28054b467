"""Tokenizers.

No tokenizer files can be fetched offline, so the default is a byte-level tokenizer (one token per
UTF-8 byte + BOS), which also pins the meaning of ``len_prompt`` in the synthetic
``conversations.json`` (traffic_generator/synth.py).  When a Hugging Face ``tokenizer.json`` is
available it is used through the ``tokenizers`` library.

Ids produced by a random-init model are arbitrary; ids outside the byte range decode to a short
printable piece so that every generated token still yields a non-empty stream line.
"""
from __future__ import annotations

import codecs
from typing import Optional


class ByteTokenizer:
    PAD, BOS, EOS = 0, 1, 2
    OFFSET = 3

    def __init__(self, vocab_size: int = 259, bos_token_id: int = 1, eos_token_id: int = 2):
        self.vocab_size = vocab_size
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id

    def encode(self, text: str, add_bos: bool = True) -> list[int]:
        ids = [b + self.OFFSET for b in text.encode("utf-8")]
        if self.vocab_size < 259:
            ids = [i % self.vocab_size for i in ids]
        return ([self.bos_token_id] if add_bos else []) + ids

    def decode(self, ids) -> str:
        d = self.stream_decoder()
        return "".join(d.step(int(i)) for i in ids) + d.flush()

    def stream_decoder(self) -> "ByteStreamDecoder":
        return ByteStreamDecoder(self)


class ByteStreamDecoder:
    """Incremental detokeniser: buffers partial UTF-8 sequences across tokens."""

    def __init__(self, tok: ByteTokenizer):
        self.tok = tok
        self.dec = codecs.getincrementaldecoder("utf-8")(errors="replace")

    def step(self, token_id: int) -> str:
        o = ByteTokenizer.OFFSET
        if o <= token_id < o + 256:
            return self.dec.decode(bytes([token_id - o]))
        if token_id in (self.tok.bos_token_id, self.tok.eos_token_id, ByteTokenizer.PAD):
            return self.dec.decode(b"", final=False)
        return self.dec.decode(b"", final=False) + f" t{token_id}"

    def flush(self) -> str:
        return self.dec.decode(b"", final=True)


class HFTokenizer:
    def __init__(self, path: str, bos_token_id: int, eos_token_id: int):
        from tokenizers import Tokenizer
        self.tk = Tokenizer.from_file(path)
        self.vocab_size = self.tk.get_vocab_size()
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id

    def encode(self, text: str, add_bos: bool = True) -> list[int]:
        ids = self.tk.encode(text, add_special_tokens=False).ids
        return ([self.bos_token_id] if add_bos else []) + ids

    def decode(self, ids) -> str:
        return self.tk.decode([int(i) for i in ids], skip_special_tokens=True)

    def stream_decoder(self):
        return _HFStreamDecoder(self)


class _HFStreamDecoder:
    """Incremental detokeniser: O(1) amortised work per token.  Only a short window of ids is ever re-decoded —
    ``ids[prefix:]``, where ``prefix`` trails the emitted text by a few tokens so that the tokenizer sees the context
    it needs for spacing / byte-fallback merges — instead of the whole sequence per token (O(n²) per stream on the
    API event loop).  Text is emitted as the difference between the window decoded with and without the new ids; a
    window ending in an incomplete UTF-8 sequence (U+FFFD) is held back until the next token completes it."""

    KEEP = 6                           # ids of already-emitted context kept in the window

    def __init__(self, tok: HFTokenizer):
        self.tok = tok
        self.ids: list[int] = []
        self.prefix = 0                # window start
        self.read = 0                  # ids[prefix:read] have been emitted

    def step(self, token_id: int) -> str:
        self.ids.append(int(token_id))
        ids = self.ids
        old = self.tok.decode(ids[self.prefix:self.read])
        new = self.tok.decode(ids[self.prefix:])
        if new.endswith("\ufffd") or len(new) < len(old):
            return ""                  # incomplete byte sequence: wait for the rest
        piece = new[len(old):]
        self.read = len(ids)
        self.prefix = max(self.read - self.KEEP, self.prefix)
        return piece

    def flush(self) -> str:
        if self.read == len(self.ids):
            return ""
        old = self.tok.decode(self.ids[self.prefix:self.read])
        new = self.tok.decode(self.ids[self.prefix:])
        self.read = len(self.ids)
        return new[len(old):] if len(new) >= len(old) else ""


def get_tokenizer(model_cfg, path: Optional[str] = None):
    if path:
        return HFTokenizer(path, model_cfg.bos_token_id, model_cfg.eos_token_id)
    return ByteTokenizer(model_cfg.vocab_size, model_cfg.bos_token_id, model_cfg.eos_token_id)
