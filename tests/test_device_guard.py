"""Multi-GPU-in-one-process safety (CPU test, no GPU calls): every native call of a handle must run with the HANDLE's
device current and on that device's stream, and must reject tensors that live elsewhere (a launch on another GPU's
stream with pointers it cannot reach is a memory fault, not an error code).  The library re-checks the current device
itself (csrc/engine.h Engine::enter -> -22)."""
import contextlib
import ctypes as C

import pytest
import torch

from pytorch_stable_diffusion_amd import _native as N


class FakeTensor:
    def __init__(self, shape, device, dtype=torch.float32):
        self.shape, self.device, self.dtype, self.is_cuda = tuple(shape), torch.device(device), dtype, True

    def dim(self):
        return len(self.shape)

    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    def contiguous(self):
        return self

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return 0x1000


class FakeLib:
    def __init__(self, log):
        self.log = log

    def __getattr__(self, name):
        def fn(*a):
            self.log.append(("call", name, self.log_state["current"]))
            return 0
        return fn


@pytest.fixture
def guarded(monkeypatch):
    log = []
    state = {"current": torch.device("cuda", 0)}

    @contextlib.contextmanager
    def fake_device(dev):
        prev = state["current"]
        state["current"] = torch.device(dev)
        log.append(("enter", torch.device(dev)))
        try:
            yield
        finally:
            state["current"] = prev

    monkeypatch.setattr(torch.cuda, "device", fake_device)
    monkeypatch.setattr(N, "cur_stream", lambda device=None: (log.append(("stream", device)), C.c_void_p(0))[1])
    monkeypatch.setattr(torch, "empty", lambda shape, dtype=None, device=None: FakeTensor(shape, device, dtype))
    lib = FakeLib(log)
    lib.log_state = state
    h = N.UNetHandle.__new__(N.UNetHandle)
    h._dev, h._h, h._lib, h.flags = torch.device("cuda", 1), C.c_void_p(1), lib, 0
    return h, log, state


def test_every_unet_entry_point_runs_on_the_handles_device(guarded):
    h, log, state = guarded
    d1 = "cuda:1"
    h.set_context(FakeTensor((2, 77, 768), d1))
    h.set_schedule(FakeTensor((50, 320), d1))
    h.forward(FakeTensor((1, 4, 64, 64), d1), 2, step_idx=0)
    h.denoise_step(FakeTensor((1, 4, 64, 64), d1), 0, True, 7.5, FakeTensor((1, 4, 64, 64), d1), (1, 1, 1, 1, 1))
    h.denoise_step(FakeTensor((4, 4, 64, 64), d1), 0, True, 7.5, FakeTensor((4, 4, 64, 64), d1), (1, 1, 1, 1, 1))   # 4 prompts, one chain
    h.ln_guard()
    h.run_block("unet.encoders.1.0", 0, FakeTensor((2, 8, 8, 320), d1), time=FakeTensor((1, 1280), d1), out_shape=(2, 8, 8, 320))
    calls = [e for e in log if e[0] == "call"]
    assert [c[1] for c in calls] == ["sdmi_unet_set_context", "sdmi_unet_set_schedule", "sdmi_unet_forward",
                                     "sdmi_unet_denoise_step_batch", "sdmi_unet_denoise_step_batch", "sdmi_unet_ln_guard",
                                     "sdmi_unet_run_block"]
    assert all(c[2] == torch.device("cuda", 1) for c in calls), calls          # issued while cuda:1 was current
    assert all(e[1] == torch.device("cuda", 1) for e in log if e[0] == "stream")   # on cuda:1's stream
    assert state["current"] == torch.device("cuda", 0)                         # and the caller's device is restored


def test_tensor_on_another_device_is_rejected_before_any_launch(guarded):
    h, log, _ = guarded
    with pytest.raises(ValueError, match="cuda:0.*cuda:1"):
        h.set_context(FakeTensor((2, 77, 768), "cuda:0"))
    with pytest.raises(ValueError):
        h.denoise_step(FakeTensor((1, 4, 64, 64), "cuda:1"), 0, True, 7.5, FakeTensor((1, 4, 64, 64), "cuda:0"), (1,) * 5)
    assert not [e for e in log if e[0] == "call"]


def test_weights_on_mixed_devices_are_rejected():
    with pytest.raises(ValueError, match="ONE cuda device"):
        N._state_device({"a": FakeTensor((1,), "cuda:0"), "b": FakeTensor((1,), "cuda:1")})
    with pytest.raises(ValueError):
        N._state_device({"a": FakeTensor((1,), "cpu")})


def test_device_spellings_do_not_drop_the_handle():
    """'cuda' and 'cuda:0' must compare equal once normalised (mixing them re-mallocs the 6 GiB arena)."""
    from pytorch_stable_diffusion_amd._util import normalize_device
    assert normalize_device("cpu") == torch.device("cpu")
    assert normalize_device("cuda:1") == torch.device("cuda", 1)
    if torch.cuda.is_available():
        assert normalize_device("cuda") == torch.device("cuda", torch.cuda.current_device())
