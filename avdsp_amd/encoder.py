"""ctypes view of avdsp_amd/lib/libavdsp_encoder.so (include/avdsp_encoder.h): the program encoder,
host-only C.  Tests drive the C API through this; `encode()` is a small convenience around
dspEncoderInit ... dsp_END_OF_CODE that returns the program words."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libavdsp_encoder.so")

_libs = {}


def lib(path: str | None = None) -> C.CDLL:
    """The encoder library; `path` selects another build with the same API (the tests load the compiled
    reference encoder this way, in the build container, to compare bytes)."""
    path = path or LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: run make -C avdsp_amd/csrc")
        L = C.CDLL(path)
        i32, f32, f64, vp = C.c_int, C.c_float, C.c_double, C.c_void_p
        L.dspEncoderInit.argtypes = [vp, i32, i32, i32, i32, i32]; L.dspEncoderInit.restype = None
        for name, args in {
            "dsp_LOAD": [i32], "dsp_STORE": [i32], "dsp_LOAD_GAIN_Fixed": [i32, f32], "dsp_GAIN_Fixed": [f32],
            "dsp_BIQUADS": [i32], "dsp_FIR": [i32], "dspBiquad_Sections": [i32], "dspFir_Delay": [i32],
            "dsp_Filter2ndOrder": [i32, f64, f64, f32], "dsp_Filter1stOrder": [i32, f64, f32],
            "dspFir_ImpulseData": [C.POINTER(f32), i32], "dsp_TPDF_CALC": [i32], "dsp_DELAY_FixedMicroSec": [i32],
            "dsp_SAT0DB_GAIN_Fixed": [f32],
        }.items():
            if hasattr(L, name):                       # the reference build lacks the extensions
                getattr(L, name).argtypes = args
        _libs[path] = L
    return _libs[path]


def encode(build, fmt: int, fmin: int, fmax: int, max_io: int = 64, capacity: int = 1 << 16,
           path: str | None = None) -> np.ndarray:
    """Run `build(L)` (a function issuing dsp_XXX calls) between dspEncoderInit and dsp_END_OF_CODE."""
    L = lib(path)
    table = np.zeros(capacity, dtype=np.uint32)
    L.dspEncoderInit(table.ctypes.data, capacity, fmt, fmin, fmax, max_io)
    build(L)
    n = L.dsp_END_OF_CODE()
    return table[:n].copy()
