"""The bundled CLIP BPE tokenizer against transformers.CLIPTokenizer on the SAME vocab/merges (CPU).

The real `vocab.json` / `merges.txt` are not available offline (the reference only links to them, data/links.txt), so the
test builds a vocabulary in the CLIP format -- the 256-symbol byte alphabet, its `</w>` variants, merges learned by a
small BPE trainer on a toy corpus, the two special tokens -- and checks id-for-id equality on prompts that exercise
case folding, whitespace, contractions, digits, punctuation runs, non-ASCII text, truncation and padding."""
import collections

import pytest

from pytorch_stable_diffusion_amd.tokenizer import BOS, EOS, CLIPTokenizer, bytes_to_unicode

CORPUS = ("a photograph of a dog running on the beach at sunset, highly detailed, sharp focus. "
          "an astronaut riding a horse on mars; oil painting in the style of the old masters! "
          "the quick brown fox jumps over the lazy dog's back 1234567890 times, it's what they'd do... "
          "cats and dogs, trees and rivers, mountains under a starry night sky -- ultra wide angle lens "
          "café naïve façade über straße 東京 の 夜 ").lower()

PROMPTS = [
    "a dog", "", "A Photograph of a DOG running on the beach", "it's the dog's ball, they'd say; we've won!",
    "  multiple   spaces\tand\nnewlines  ", "1234 apples & 56 oranges... (really?!)", "café naïve über straße",
    "東京の夜景 neon lights", "highly-detailed, 8k --ar 16:9 <|endoftext|> trailing", "unseenword zzzqqq xylophone",
    " ".join(["a very long prompt about dogs and cats"] * 20),
]


def _train(n_merges=300):
    byte_chars = list(bytes_to_unicode().values())
    words = collections.Counter()
    enc = bytes_to_unicode()
    import regex as re
    pat = re.compile(r"'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+")
    for w in pat.findall(CORPUS):
        sym = [enc[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for i in range(len(w) - 1):
                pairs[(w[i], w[i + 1])] += c
        if not pairs:
            break
        (a, b), _cnt = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        merges.append((a, b))
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and w[i] == a and w[i + 1] == b:
                    out.append(a + b); i += 2
                else:
                    out.append(w[i]); i += 1
            new[tuple(out)] += c
        words = new
    vocab = {}
    for ch in byte_chars:
        vocab[ch] = len(vocab)
    for ch in byte_chars:
        vocab[ch + "</w>"] = len(vocab)
    for a, b in merges:
        vocab.setdefault(a + b, len(vocab))
    vocab[BOS] = len(vocab)
    vocab[EOS] = len(vocab)
    return vocab, merges


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    import json
    vocab, merges = _train()
    d = tmp_path_factory.mktemp("clip_tok")
    vf, mf = d / "vocab.json", d / "merges.txt"
    vf.write_text(json.dumps(vocab, ensure_ascii=False), encoding="utf-8")
    mf.write_text("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n", encoding="utf-8")
    return str(vf), str(mf), vocab, merges


def test_matches_transformers_clip_tokenizer(files):
    transformers = pytest.importorskip("transformers")
    vf, mf, vocab, merges = files
    ours = CLIPTokenizer(vf, merges_file=mf)                      # the notebook's constructor call
    try:
        theirs = transformers.CLIPTokenizer(vocab=vocab, merges=[(a, b) for a, b in merges])     # transformers >= 5
    except (TypeError, ValueError):
        theirs = transformers.CLIPTokenizer(vf, merges_file=mf)                                  # transformers 4.x
    for text in PROMPTS:
        got = ours.batch_encode_plus([text], padding="max_length", max_length=77).input_ids[0]
        ref = theirs([text], padding="max_length", max_length=77, truncation=True)["input_ids"][0]
        assert len(got) == 77
        assert got == list(ref), text


def test_call_surface_of_generate(files):
    vf, mf, vocab, _ = files
    tok = CLIPTokenizer(vf, merges_file=mf)
    enc = tok.batch_encode_plus(["a dog", "an astronaut riding a horse"], padding="max_length", max_length=77)
    assert len(enc.input_ids) == 2 and all(len(r) == 77 for r in enc.input_ids)
    assert enc.input_ids[0][0] == vocab[BOS] and enc["input_ids"][0].count(vocab[EOS]) >= 1
    assert enc.attention_mask[0][:4] == [1, 1, 1, 1] and enc.attention_mask[0][-1] == 0
    # cached and uncached BPE agree; unknown symbols cannot occur (byte-level alphabet is complete)
    assert tok.tokenize_ids("dog dog") == tok.tokenize_ids("dog") * 2
    with pytest.raises(TypeError):
        tok.batch_encode_plus("a dog")
    with pytest.raises(ValueError):
        CLIPTokenizer(vf)
