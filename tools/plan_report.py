"""Per-shape GEMM plan table of one UNet step (SDMI_TUNE_LOG=1): isolated tuned time x calls per forward."""
import os, sys
os.environ["SDMI_TUNE_LOG"] = "1"
import torch
sys.path.insert(0, ".")
from pytorch_stable_diffusion_amd import arch, synth, _native as N
sd = {k: v.cuda() for k, v in synth.synth_state_dict(arch.diffusion_manifest(), torch.float16).items()}
h = N.UNetHandle(sd, N.FLAG_STREAM_F32)
g = torch.Generator().manual_seed(0)
h.set_context(torch.randn((2, 77, 768), generator=g).cuda()); h.set_schedule(torch.randn((1, 320), generator=g).cuda())
lat = torch.randn((1, 4, 64, 64), generator=g).cuda()
h.forward(lat, 2, step_idx=0)
torch.cuda.synchronize()
h.close()
