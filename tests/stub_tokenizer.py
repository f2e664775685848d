"""The stub tokenizer lives in the package (pytorch_stable_diffusion_amd/tokenizer.py); kept here for the tests' imports."""
from pytorch_stable_diffusion_amd.tokenizer import StubTokenizer  # noqa: F401
