#!/usr/bin/env python3
"""Golden fixture for the checkpoint converter: EXECUTE the reference's model_converter on symbolic tensors
(build container only) and record, for every destination key, which checkpoint keys feed it and through which
op (copy / cat(dim 0) / reshape).  Output: converter_map.json (data: key names and ops, no code)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/sd")

import torch  # noqa: E402


class Sym:
    def __init__(self, op, src, shape=None):
        self.op, self.src, self.shape = op, src, shape

    def reshape(self, shape):
        return Sym(self.op + "+reshape", self.src, list(shape))

    def to_json(self):
        d = {"op": self.op, "src": self.src}
        if self.shape is not None:
            d["shape"] = self.shape
        return d


class RecordingDict(dict):
    def __getitem__(self, k):
        return Sym("copy", [k])


def main():
    real_load, real_cat = torch.load, torch.cat
    torch.load = lambda *a, **k: {"state_dict": RecordingDict()}

    def sym_cat(seq, dim=0):
        assert dim == 0
        src = []
        for s in seq:
            assert s.op == "copy"
            src += s.src
        return Sym("cat", src)

    torch.cat = sym_cat
    try:
        import model_converter
        out = model_converter.load_from_standard_weights("unused.ckpt", "cpu")
    finally:
        torch.load, torch.cat = real_load, real_cat
    fixture = {name: {k: v.to_json() for k, v in d.items()} for name, d in out.items()}
    with open(os.path.join(HERE, "converter_map.json"), "w") as f:
        json.dump(fixture, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in fixture.items()})


if __name__ == "__main__":
    main()
