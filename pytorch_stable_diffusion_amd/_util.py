"""Small host-side helpers shared by the model wrappers."""
from __future__ import annotations

import torch


def normalize_device(device) -> torch.device:
    """``torch.device`` with an explicit index for cuda ('cuda' -> 'cuda:<current>'), so that 'cuda' and 'cuda:0'
    compare equal and a model is not re-packed (arena re-allocated, plans re-read) when the two spellings mix."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())
    return device
