"""Duck-typed stand-in for CLIPTokenizer (the real vocab/merges files are not available offline):
``batch_encode_plus([text], padding="max_length", max_length=77).input_ids`` like sd/pipeline.py:109."""
import zlib


class _Enc:
    def __init__(self, ids):
        self.input_ids = ids


class StubTokenizer:
    BOS, EOS = 49406, 49407

    def batch_encode_plus(self, texts, padding=None, max_length=77):
        out = []
        for t in texts:
            words = [320 + (zlib.crc32(w.encode()) % 40000) for w in t.split()][: max_length - 2]
            ids = [self.BOS] + words + [self.EOS]
            ids += [self.EOS] * (max_length - len(ids))
            out.append(ids)
        return _Enc(out)
