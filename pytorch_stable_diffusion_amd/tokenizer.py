"""CLIP byte-level BPE tokenizer with the call surface `generate()` uses.

The reference does not ship a tokenizer: its notebook builds `transformers.CLIPTokenizer("../data/vocab.json",
merges_file="../data/merges.txt")` (sd/inference_demo.ipynb:47) and `sd/pipeline.py:109,115,127` only ever calls
`tokenizer.batch_encode_plus([text], padding="max_length", max_length=77).input_ids`.  Recent transformers releases no
longer expose `batch_encode_plus` on that class, so the drop-in needs its own reader of the same two files.

Algorithm (OpenAI CLIP `simple_tokenizer`, as published): NFC-normalise, collapse whitespace, lower-case; split with
the CLIP pattern; map each piece's UTF-8 bytes to the printable byte alphabet; merge pairs by rank with the last symbol
carrying the `</w>` end-of-word mark; look the symbols up in `vocab.json`; wrap in <|startoftext|> ... <|endoftext|> and
pad with <|endoftext|> (the SD v1 tokenizer's pad token).  Pinned in `tests/test_tokenizer.py` against
`transformers.CLIPTokenizer` on the same vocab/merges files.
"""
from __future__ import annotations

import json
import unicodedata
from functools import lru_cache
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import regex as re

BOS = "<|startoftext|>"
EOS = "<|endoftext|>"

_PATTERN = re.compile(
    r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", re.IGNORECASE)


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """The reversible byte -> printable-character table of GPT-2 / CLIP byte-level BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


class Encoding(dict):
    """Minimal stand-in for transformers' BatchEncoding: attribute and key access to `input_ids`/`attention_mask`."""

    def __getattr__(self, item):
        try:
            return self[item]
        except KeyError as exc:
            raise AttributeError(item) from exc


class CLIPTokenizer:
    def __init__(self, vocab_file: Union[str, Dict[str, int]], merges_file: Union[str, Sequence[str], None] = None,
                 *, merges: Union[str, Sequence[str], None] = None, pad_token: str = EOS):
        merges_src = merges_file if merges_file is not None else merges
        if merges_src is None:
            raise ValueError("CLIPTokenizer needs both the vocabulary and the merges file")
        if isinstance(vocab_file, dict):
            self.encoder = dict(vocab_file)
        else:
            with open(vocab_file, encoding="utf-8") as f:
                self.encoder = json.load(f)
        if isinstance(merges_src, str):
            with open(merges_src, encoding="utf-8") as f:
                lines = f.read().split("\n")
        else:
            lines = list(merges_src)
        pairs: List[Tuple[str, str]] = []
        for ln in lines:
            if not ln or ln.startswith("#version"):
                continue
            a, b = ln.split()
            pairs.append((a, b))
        self.bpe_ranks = {p: i for i, p in enumerate(pairs)}
        for tok in (BOS, EOS, pad_token):
            if tok not in self.encoder:
                raise KeyError(f"vocabulary has no {tok!r} entry")
        self.bos_token_id = self.encoder[BOS]
        self.eos_token_id = self.encoder[EOS]
        self.pad_token_id = self.encoder[pad_token]
        self.byte_encoder = bytes_to_unicode()
        self._cache: Dict[str, Tuple[str, ...]] = {}

    # ---- BPE ------------------------------------------------------------------------------------
    def _bpe(self, token: str) -> Tuple[str, ...]:
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        inf = float("inf")
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            first, second = min(pairs, key=lambda pr: self.bpe_ranks.get(pr, inf))
            if (first, second) not in self.bpe_ranks:
                break
            out: List[str] = []
            i = 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    out.append(first + second)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        self._cache[token] = word
        return word

    # ---- text -> ids ------------------------------------------------------------------------------
    @staticmethod
    def _normalise(text: str) -> str:
        text = unicodedata.normalize("NFC", text)
        text = re.sub(r"\s+", " ", text)
        return text.lower()

    def tokenize_ids(self, text: str) -> List[int]:
        ids: List[int] = []
        unk = self.encoder[EOS]
        for piece in _PATTERN.findall(self._normalise(text)):
            if piece in (BOS, EOS):
                ids.append(self.encoder[piece])
                continue
            sym = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder.get(s, unk) for s in self._bpe(sym))
        return ids

    def encode(self, text: str, max_length: Optional[int] = None, padding: Union[bool, str] = False,
               truncation: bool = True) -> List[int]:
        ids = self.tokenize_ids(text)
        if max_length is not None and truncation and len(ids) > max_length - 2:
            ids = ids[:max_length - 2]
        ids = [self.bos_token_id] + ids + [self.eos_token_id]
        if padding == "max_length" and max_length is not None:
            ids = ids + [self.pad_token_id] * (max_length - len(ids))
        return ids

    def batch_encode_plus(self, batch_text: Iterable[str], padding: Union[bool, str] = False,
                          max_length: Optional[int] = None, truncation: bool = True, **_unused) -> Encoding:
        """sd/pipeline.py:109: `.batch_encode_plus([prompt], padding="max_length", max_length=77).input_ids`."""
        if isinstance(batch_text, str):
            raise TypeError("batch_encode_plus takes a list of strings")
        rows = [self.encode(t, max_length=max_length, padding=padding, truncation=truncation) for t in batch_text]
        if padding is True or padding == "longest":
            width = max(len(r) for r in rows)
            rows = [r + [self.pad_token_id] * (width - len(r)) for r in rows]
        mask = []
        for r in rows:                      # 1 up to and including the first <|endoftext|>, 0 on the padding
            n = r.index(self.eos_token_id) + 1 if self.eos_token_id in r else len(r)
            mask.append([1] * n + [0] * (len(r) - n))
        return Encoding(input_ids=rows, attention_mask=mask)

    __call__ = batch_encode_plus


class StubTokenizer:
    """Duck-typed stand-in for CLIPTokenizer when vocab.json / merges.txt are not available (benchmarks, tests, the
    replicas launcher's --stub-tokenizer): ``batch_encode_plus([text], padding="max_length", max_length=77).input_ids``
    as sd/pipeline.py:109 calls it; word ids are CRC32 hashes, so equal prompts give equal ids in every process."""
    BOS, EOS = 49406, 49407

    def batch_encode_plus(self, texts, padding=None, max_length=77, **_unused) -> Encoding:
        import zlib
        rows = []
        for t in texts:
            words = [320 + (zlib.crc32(w.encode()) % 40000) for w in t.split()][: max_length - 2]
            ids = [self.BOS] + words + [self.EOS]
            rows.append(ids + [self.EOS] * (max_length - len(ids)))
        return Encoding(input_ids=rows, attention_mask=[[1] * len(r) for r in rows])
