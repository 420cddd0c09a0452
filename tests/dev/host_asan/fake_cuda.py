"""pytest plugin for tests/dev/host_asan/run.sh only: lets the `-m gpu` tests that hand DEVICE pointers to the library run their host
side on a machine without a GPU -- "cuda" tensors are CPU tensors, streams are 0.  The stand-in device layer ignores device pointers;
the tests' sample comparisons fail, which is expected there (run.sh looks for sanitizer reports, nothing else)."""
import functools
import types

import torch


def _cpu_device(fn):
    @functools.wraps(fn)
    def wrapped(*a, **kw):
        if str(kw.get("device", "")).startswith("cuda"):
            kw["device"] = "cpu"
        kw.pop("pin_memory", None)
        return fn(*a, **kw)
    return wrapped


for _name in ("zeros", "empty", "ones", "full", "tensor", "arange", "randn", "rand", "randint", "zeros_like", "empty_like", "as_tensor"):
    setattr(torch, _name, _cpu_device(getattr(torch, _name)))
torch.Tensor.cuda = lambda self, *a, **kw: self
_to = torch.Tensor.to
torch.Tensor.to = lambda self, *a, **kw: self if (a and str(a[0]).startswith("cuda")) or str(kw.get("device", "")).startswith("cuda") else _to(self, *a, **kw)
torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 1
torch.cuda.synchronize = lambda *a, **kw: None
torch.cuda.set_device = lambda *a, **kw: None
torch.cuda.current_stream = lambda *a, **kw: types.SimpleNamespace(cuda_stream=0, synchronize=lambda: None)
torch.cuda.Stream = lambda *a, **kw: types.SimpleNamespace(cuda_stream=0, synchronize=lambda: None, wait_stream=lambda s: None)
