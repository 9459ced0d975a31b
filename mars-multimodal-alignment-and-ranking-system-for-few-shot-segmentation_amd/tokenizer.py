"""Byte-pair tokenizer of the CLIP / AlphaCLIP text towers (host side, pure Python).

Follows /root/reference/clip/simple_tokenizer.py:15-127 (byte alphabet, merge ranks, the
lower-cased regex pre-split, greedy lowest-rank merging with the ``</w>`` word-end mark) and
``tokenize`` of /root/reference/clip/clip.py:205-245 and
/root/reference/alpha_clip/alpha_clip.py:210-250 (start / end tokens, zero padding to the
context length, int32 output, error or truncation for long prompts).

The merge table (``bpe_simple_vocab_16e6.txt.gz``, shipped with CLIP) is data the user supplies:
pass its path, or set ``MARS_BPE_PATH``.  ``ftfy.fix_text`` of the reference is the identity on
the ASCII prompts the pipeline builds ('a {name}, {description}.'); non-ASCII prompts are
NFC-normalised instead, which is what ftfy does to already well-formed text.
"""
from __future__ import annotations

import gzip
import html
import os
import unicodedata
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import regex
import torch

N_MERGES = 49152 - 256 - 2          # merges actually used (simple_tokenizer.py:67)
SOT, EOT = "<|startoftext|>", "<|endoftext|>"
_SPLIT = regex.compile(
    r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
_SPACES = regex.compile(r"\s+")


def byte_alphabet() -> List[str]:
    """Printable stand-in for each of the 256 byte values: the printable Latin-1 ranges map to
    themselves, the remaining bytes to code points 256, 257, ... in byte order."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, spare = [], 0
    for b in range(256):
        if b in keep:
            table.append(chr(b))
        else:
            table.append(chr(256 + spare))
            spare += 1
    return table


class BpeTokenizer:
    def __init__(self, bpe_path: str | None = None):
        bpe_path = bpe_path or os.environ.get("MARS_BPE_PATH")
        if not bpe_path or not os.path.exists(bpe_path):
            raise FileNotFoundError(
                "the CLIP merge table (bpe_simple_vocab_16e6.txt.gz) is needed to tokenise arbitrary prompts: "
                "pass bpe_path= or set MARS_BPE_PATH (prompts in synth.PROMPT_TOKENS work without it)")
        with gzip.open(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges: List[Tuple[str, str]] = [tuple(l.split()) for l in lines[1:1 + N_MERGES]]  # line 0 is a version header
        self.byte_sym = byte_alphabet()
        # vocabulary order: the 256 byte symbols as in simple_tokenizer.py:26-35 (kept bytes first in
        # value order, then the re-mapped ones), the same with the word-end mark, merges, specials
        order = [b for b in range(256) if self.byte_sym[b] == chr(b)] + [b for b in range(256) if self.byte_sym[b] != chr(b)]
        base = [self.byte_sym[b] for b in order]
        vocab = base + [s + "</w>" for s in base] + ["".join(m) for m in merges] + [SOT, EOT]
        self.encoder: Dict[str, int] = {s: i for i, s in enumerate(vocab)}
        self.decoder: Dict[int, str] = {i: s for s, i in self.encoder.items()}
        self.rank: Dict[Tuple[str, str], int] = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.encoder[SOT], self.encoder[EOT]
        self._memo: Dict[str, List[str]] = {SOT: [SOT], EOT: [EOT]}

    # ------------------------------------------------------------------ one pre-split token
    def _merge_word(self, token: str) -> List[str]:
        hit = self._memo.get(token)
        if hit is not None:
            return hit
        parts = list(token[:-1]) + [token[-1] + "</w>"]
        while len(parts) > 1:
            # the adjacent pair that was learnt first
            best, best_rank = -1, None
            for i in range(len(parts) - 1):
                r = self.rank.get((parts[i], parts[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best < 0:
                break
            a, b = parts[best], parts[best + 1]
            # every non-overlapping occurrence of that pair, left to right
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self._memo[token] = parts
        return parts

    # ------------------------------------------------------------------ text -> ids
    @staticmethod
    def clean(text: str) -> str:
        text = unicodedata.normalize("NFC", text)
        text = html.unescape(html.unescape(text)).strip()
        return _SPACES.sub(" ", text).strip().lower()

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for piece in _SPLIT.findall(self.clean(text)):
            sym = "".join(self.byte_sym[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._merge_word(sym))
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        inv = {s: b for b, s in enumerate(self.byte_sym)}
        text = "".join(self.decoder[int(i)] for i in ids)
        special = text.replace(SOT, "").replace(EOT, "")
        raw = bytearray()
        for ch in special.replace("</w>", " "):
            raw.append(inv[ch] if ch in inv else ord(" "))
        return raw.decode("utf-8", errors="replace")

    def __call__(self, texts: Union[str, Sequence[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.int32)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, : len(ids)] = torch.tensor(ids, dtype=torch.int32)
        return out


def default_tokenizer(bpe_path: str | None = None):
    """The merge-table tokenizer when the table is available, else the bundled prompt table."""
    try:
        return BpeTokenizer(bpe_path)
    except FileNotFoundError:
        from .synth import tokenize
        return tokenize
