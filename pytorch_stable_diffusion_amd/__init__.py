"""MI355X-native Stable Diffusion denoising path (drop-in for dawmro/pytorch_stable_diffusion's
``pipeline.generate`` / ``models[...]`` surface).  See DESIGN.md."""
__version__ = "0.1.0"
