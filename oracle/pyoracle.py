"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  The
product package (avdsp_amd/) never does; it fails loudly when its HIP library is missing instead
of falling back to this CPU code.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")


def build(force: bool = False) -> str:
    """Compile liboracle.so (gcc, seconds).  Also builds oracle/_ref when /root/reference exists."""
    src = [os.path.join(HERE, f) for f in ("avdsp_oracle.c", "oracle_interp.inc", "avdsp_oracle.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, i32, f32, f64, i64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_longlong
        L.oracle_new.restype = vp; L.oracle_new.argtypes = [i32]
        L.oracle_free.argtypes = [vp]
        L.oracle_init.restype = i32; L.oracle_init.argtypes = [vp, vp, i32, i32, i32, i32]
        L.oracle_reset.restype = i32; L.oracle_reset.argtypes = [vp, i32, i32, i32]
        L.oracle_find_core.restype = vp; L.oracle_find_core.argtypes = [vp, i32]
        L.oracle_find_core_begin.restype = vp; L.oracle_find_core_begin.argtypes = [vp]
        L.oracle_run.restype = i32; L.oracle_run.argtypes = [vp, vp, vp, vp]
        L.oracle_run_block.restype = i32
        L.oracle_run_block.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, i32, i32, i32]
        L.oracle_run_block_frame.restype = i32
        L.oracle_run_block_frame.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, i32, i32, vp]
        L.oracle_mul_float_double.restype = f64; L.oracle_mul_float_double.argtypes = [f32, f32]
        L.oracle_mul_float_float.restype = f32; L.oracle_mul_float_float.argtypes = [f32, f32]
        L.oracle_int_to_float_scaled.restype = f32; L.oracle_int_to_float_scaled.argtypes = [i32, i32]
        L.oracle_int_to_double_scaled.restype = f64; L.oracle_int_to_double_scaled.argtypes = [i32, i32]
        L.oracle_s31_from_double.restype = i32; L.oracle_s31_from_double.argtypes = [f64]
        L.oracle_s31_from_float.restype = i32; L.oracle_s31_from_float.argtypes = [f32]
        L.oracle_saturate_double.restype = f64; L.oracle_saturate_double.argtypes = [f64]
        L.oracle_saturate_float.restype = f32; L.oracle_saturate_float.argtypes = [f32]
        L.oracle_truncate_double.restype = f64; L.oracle_truncate_double.argtypes = [f64, i32]
        L.oracle_truncate_float.restype = f32; L.oracle_truncate_float.argtypes = [f32, i32]
        L.oracle_saturate64_031.restype = i64; L.oracle_saturate64_031.argtypes = [i64, i32]
        L.oracle_biquads_int.restype = i64; L.oracle_biquads_int.argtypes = [i32, vp, vp, i32, i32]
        L.oracle_biquads_double.restype = f64; L.oracle_biquads_double.argtypes = [f32, vp, vp, i32, i32]
        L.oracle_fir_double.restype = f64; L.oracle_fir_double.argtypes = [f32, vp, vp, i32]
        L.oracle_qnm.restype = i64; L.oracle_qnm.argtypes = [f64, i32, i32]
        _lib = L
    return _lib


def sample_dtype(fmt: int):
    return np.float32 if fmt in (5, 6) else np.int32


class OracleProgram:
    """One loaded program: the caller-visible shape of dspRuntimeInit + dspRuntime_N on the CPU."""

    def __init__(self, fmt: int, prog_words: np.ndarray, fs: int = 48000, random: int = 0,
                 dither: int = 31, max_size: int | None = None, after: "OracleProgram | None" = None):
        """`after`: load this program into the context another program was loaded into before (the reference keeps its
        rate count and dither state in statics that survive a dspRuntimeInit); that program must not be used afterwards."""
        self.L = lib()
        self.fmt = fmt
        n = int(prog_words[1]) + max(int(np.int32(prog_words[2])), 0)
        self.buf = np.zeros(max(n, len(prog_words)) + 64, dtype=np.uint32)
        self.buf[:len(prog_words)] = prog_words
        if after is not None:
            assert after.fmt == fmt
            self.ctx, after.ctx = after.ctx, None
        else:
            self.ctx = self.L.oracle_new(fmt)
        if not self.ctx:
            raise ValueError(f"unsupported DSP_FORMAT {fmt}")
        self.rc = self.L.oracle_init(self.ctx, self.buf.ctypes.data, n if max_size is None else max_size,
                                     fs, random, dither)
        self.cores = []
        if self.rc >= 0:
            self.data_ptr = self.buf.ctypes.data + 4 * self.rc
            k = 1
            while True:
                p = self.L.oracle_find_core(self.buf.ctypes.data, k)
                if not p:
                    break
                self.cores.append(self.L.oracle_find_core_begin(p))
                k += 1

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.L.oracle_free(self.ctx)
                self.ctx = None
        except Exception:
            pass

    @property
    def state(self) -> np.ndarray:
        return self.buf[self.rc:self.rc + int(self.buf[2])]

    def reset(self, fs: int, random: int = 0, dither: int = 31) -> int:
        return self.L.oracle_reset(self.ctx, fs, random, dither)

    def run_block(self, x: np.ndarray, out_stride: int, in_io_base: int, out_io_base: int = 0,
                  scratch_len: int | None = None, block: int | None = None,
                  out: np.ndarray | None = None, frame: np.ndarray | None = None) -> np.ndarray:
        """Host loop of linux/avdsp_plugin.c:95-142: per block, cores outer, frames inner.
        `frame` (uint32, at least the IO span): the samples[] array kept by the caller, so that slots
        outside the two windows persist from frame to frame, between cores and between calls; without
        it every call starts from a zeroed frame."""
        x = np.ascontiguousarray(x, dtype=sample_dtype(self.fmt))
        nframes, in_stride = x.shape
        if out is None:
            out = np.zeros((nframes, out_stride), dtype=sample_dtype(self.fmt))
        if scratch_len is None:
            scratch_len = max(in_io_base + in_stride, out_io_base + out_stride) + 1
        block = block or nframes
        for b0 in range(0, nframes, block):
            b1 = min(b0 + block, nframes)
            for core in self.cores:
                if frame is not None:
                    assert frame.dtype == np.uint32 and frame.flags.c_contiguous
                    self.L.oracle_run_block_frame(self.ctx, core, self.data_ptr,
                                                  x[b0:b1].ctypes.data, in_stride, in_io_base,
                                                  out[b0:b1].ctypes.data, out_stride, out_io_base,
                                                  b1 - b0, frame.ctypes.data)
                    continue
                self.L.oracle_run_block(self.ctx, core, self.data_ptr,
                                        x[b0:b1].ctypes.data, in_stride, in_io_base,
                                        out[b0:b1].ctypes.data, out_stride, out_io_base,
                                        b1 - b0, scratch_len)
        return out


def have_ref() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "ref_driver"))


def run_reference(fmt: int, prog_words: np.ndarray, x: np.ndarray, out_stride: int, in_io_base: int,
                  out_io_base: int = 0, fs: int = 48000, random: int = 0, dither: int = 31,
                  block: int | None = None, scratch_len: int | None = None, max_size: int = 0,
                  tmpdir: str = "/tmp", want_state: bool = False):
    """Run the COMPILED REFERENCE (oracle/_ref, build container only) on the same program/input.

    Returns (init_rc, out [, buffer_after]).  Runs in a child process (ref_driver): the reference
    libraries are built -Ofast and switch the loading thread to flush-to-zero."""
    import tempfile
    x = np.ascontiguousarray(x, dtype=sample_dtype(fmt))
    nframes, in_stride = x.shape
    if scratch_len is None:
        scratch_len = max(in_io_base + in_stride, out_io_base + out_stride) + 1
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        pp, ip, op, sp = (os.path.join(d, n) for n in ("p.bin", "in.raw", "out.raw", "state.raw"))
        np.asarray(prog_words, dtype=np.uint32).tofile(pp)
        x.tofile(ip)
        cmd = [os.path.join(REF_DIR, "ref_driver"), os.path.join(REF_DIR, f"libavdspref_{fmt}.so"),
               str(fmt), pp, str(max_size), str(fs), str(random), str(dither), ip, op,
               str(nframes), str(block or nframes), str(in_stride), str(in_io_base),
               str(out_stride), str(out_io_base), str(scratch_len), sp]
        res = subprocess.run(cmd, capture_output=True, text=True, check=True)
        rc = int(res.stdout.split("init=")[1].split()[0])
        if rc < 0:
            return (rc, None, None) if want_state else (rc, None)
        out = np.fromfile(op, dtype=sample_dtype(fmt)).reshape(nframes, out_stride)
        if want_state:
            return rc, out, np.fromfile(sp, dtype=np.uint32)
        return rc, out
