#!/usr/bin/env python3
"""What fp16 STORAGE alone costs under the stress weight law (synth.py law="stress"), operand class by operand class.

The oracle (oracle/unet_ref.py, fp32 arithmetic throughout, pinned against the reference) is run with ONE class of operands
rounded to fp16 at a time -- the weights of every conv / linear ("w"), the activation operand of every conv / linear ("a"),
q / k / v / probabilities of the attentions ("attn") -- through the reference's own 20-step txt2img loop
(tests/golden/stress_e2e.npz: the image the imported reference produced on this law; same seed, same noise stream, benign VAE
decoder through the oracle).  The pixel MAE of each run against that golden is the FLOOR of any implementation that stores that
operand class in fp16 and does everything else exactly: north_star's 1e-3 is reachable under this law only if the floors of the
classes the path rounds stay below it.  Result: tests/golden/stress_floor.json (committed; asserted by tests/test_oracle_full.py).

  python tests/golden/stress_floor.py [steps=20]
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import aux_ref, ddpm_ref, unet_ref  # noqa: E402
from pytorch_stable_diffusion_amd import arch, model_loader, synth  # noqa: E402
from pytorch_stable_diffusion_amd.tokenizer import StubTokenizer  # noqa: E402


@torch.no_grad()
def run(sd, aux, ctx, steps, quant):
    unet_ref.QUANT = set(quant)
    try:
        sched = ddpm_ref.RefSchedule()
        sched.set_inference_timesteps(steps)
        g = torch.Generator().manual_seed(42)
        lat = torch.randn((1, 4, 64, 64), generator=g)
        lat = ddpm_ref.denoise_loop(lambda x, c, t: unet_ref.diffusion_forward(sd, x, c, t), lat, ctx, sched, g)
    finally:
        unet_ref.QUANT = set()
    img = aux_ref.vae_decode(aux["decoder"], lat)                      # (1, 3, 512, 512) in [-1, 1]
    u8 = ((img.clamp(-1, 1) + 1) * 127.5).permute(0, 2, 3, 1)[0].to(torch.uint8).numpy()
    return u8, img[0]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    gold = np.load(os.path.join(HERE, "stress_e2e.npz" if steps == 20 else f"stress_e2e{steps}.npz"))
    ref_u8 = gold["txt20_u8"] if steps == 20 else gold["u8"]
    sd = synth.synth_state_dict(arch.diffusion_manifest(), law="stress")
    aux = model_loader.synthetic_state_dicts(("clip", "decoder"))
    tok = StubTokenizer()
    ids = lambda t: torch.tensor(tok.batch_encode_plus([t], padding="max_length", max_length=77).input_ids)
    ctx = torch.cat([aux_ref.clip_forward(aux["clip"], ids("a dog")), aux_ref.clip_forward(aux["clip"], ids(""))])
    out = {"steps": steps, "threads": torch.get_num_threads(), "torch": torch.__version__, "runs": {}}
    for name, quant in (("fp32 (oracle itself)", ()), ("weights fp16", ("w",)), ("activation operands fp16", ("a",)),
                        ("attention q/k/v/p fp16", ("attn",)), ("weights + activations + attention fp16", ("w", "a", "attn"))):
        t0 = time.time()
        u8, _ = run(sd, aux, ctx, steps, quant)
        mae = float(np.abs(u8.astype(np.float64) - ref_u8.astype(np.float64)).mean() / 255.0)
        mx = int(np.abs(u8.astype(np.int32) - ref_u8.astype(np.int32)).max())
        out["runs"][name] = {"quant": list(quant), "pixel_mae": mae, "u8_max_diff": mx}
        print(f"{name:45s} pixel MAE {mae:.3e}  uint8 max diff {mx}   ({time.time() - t0:.0f} s)", flush=True)
        json.dump(out, open(os.path.join(HERE, "stress_floor.json" if steps == 20 else f"stress_floor{steps}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
