"""Child process of test_plan_cache_*: builds a partial UNet handle, runs two blocks, prints
``<tuned_shapes> <sha1 of the outputs>``.  Plans come from SDMI_PLAN_FILE / SDMI_PLAN_CACHE_DIR (engine.h PlanStore)."""
import hashlib
import sys

import torch

from pytorch_stable_diffusion_amd import _native as N
from tests import helpers as H


def main():
    dev = "cuda"
    meta = H.blocks_meta()["blocks"]
    names = ["res_320_640", "attn_8_80"]
    state = {}
    for n in names:
        for k, v in H.block_weights(meta[n]["prefix"]).items():
            state[k] = v.to(dev)
    h = N.UNetHandle(state, N.FLAG_PARTIAL | N.FLAG_STREAM_F32)
    h.set_context(H.seeded((2, 77, 768), 7).to(dev))
    sha = hashlib.sha1()
    for n in names:
        m = meta[n]
        x = H.seeded(tuple(m["ishape"]), m["seed"]).permute(0, 2, 3, 1).contiguous().to(dev)
        kind = {"res": 0, "attn": 1}[m["kind"]]
        time = H.seeded((1, 1280), 8).to(dev) if kind == 0 else None
        co = m["args"][1] if kind == 0 else x.shape[3]
        out = h.run_block(m["prefix"], kind, x, time=time, out_shape=(x.shape[0], x.shape[1], x.shape[2], co))
        torch.cuda.synchronize()
        sha.update(out.cpu().numpy().tobytes())
    print(h.tuned_shapes, sha.hexdigest())
    h.close()


if __name__ == "__main__":
    sys.exit(main())
