#!/usr/bin/env python3
"""What fp16 STORAGE alone costs ONE UNet forward under the stress weight law (64x64 latents, t = 980, the inputs of
tests/golden/stress.npz "unet_64_t980"): the fp32 oracle with one operand class rounded to fp16 at a time, rel-L2 against the
imported reference's own output.  The floors the single-forward tests assert against (tests/test_gpu_stress.py): the default
mode rounds all three classes, the accurate mode (SDMI_FLAG_ACCURATE) only the weights and the attention operands.

  python tests/golden/stress_floor_forward.py      -> tests/golden/stress_floor_forward.json
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ddpm_ref, unet_ref  # noqa: E402
from pytorch_stable_diffusion_amd import arch, synth  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    sd = synth.synth_state_dict(arch.diffusion_manifest(), law="stress")
    ref = torch.from_numpy(H.load_npz("stress.npz")["unet_64_t980"])
    lat = H.seeded((1, 4, 64, 64), 0).repeat(2, 1, 1, 1)
    ctx = H.seeded((2, 77, 768), 1)
    t = ddpm_ref.time_embedding(980)
    out = {"threads": torch.get_num_threads(), "torch": torch.__version__, "runs": {}}
    for name, q in (("fp32 (oracle itself)", ()), ("weights fp16", ("w",)), ("activation operands fp16", ("a",)),
                    ("weights + attention fp16", ("w", "attn")), ("weights + activations + attention fp16", ("w", "a", "attn"))):
        unet_ref.QUANT = set(q)
        try:
            with torch.no_grad():
                got = unet_ref.diffusion_forward(sd, lat, ctx, t)
        finally:
            unet_ref.QUANT = set()
        out["runs"][name] = {"rel_l2": H.rel_l2(got, ref)}
        print(name, out["runs"][name], flush=True)
    with open(os.path.join(HERE, "stress_floor_forward.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
