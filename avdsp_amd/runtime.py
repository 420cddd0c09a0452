"""Python mirror of the AVDSP runtime interface over the C-ABI library libavdsp_mi355x.so.

Same names, argument meaning and error behaviour as the reference's C API
(module_avdsp/runtime/dsp_runtime.h:160-164; host call sequence linux/avdsp_plugin.c:309-356,
linux/dsprun.c:85-132): dspRuntimeInit -> dspFindCore/dspFindCoreBegin -> dspRuntimeReset ->
dspRuntime_N per frame, plus the block extension declared in include/avdsp_runtime.h.

There is no CPU execution path here.  If the HIP library has not been built, or no GPU is visible,
the calls raise / return the library's negative error codes -- they never fall back to the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVDSP_LIB") or os.path.join(HERE, "lib", "libavdsp_mi355x.so")
CSRC = os.path.join(HERE, "csrc")

EXPORTED = [
    # reference API
    "dspFindCore", "dspFindCoreBegin", "dspRuntimeReset", "dspRuntimeInit",
    "dspRuntime_2", "dspRuntime_3", "dspRuntime_4", "dspRuntime_5", "dspRuntime_6",
    "dspHeaderPtr", "dspBiquadFreqSkip", "dspMantissa", "dspOpcodeText", "dspQNM", "dspQM64", "dspQM32",
    # block extension (include/avdsp_runtime.h)
    "dspRuntimeBlock_2", "dspRuntimeBlock_3", "dspRuntimeBlock_4", "dspRuntimeBlock_5", "dspRuntimeBlock_6",
    "dspRuntimeStrandInfo", "dspRuntimeBlockSubmit", "dspRuntimeBlockWait", "dspRuntimeBlockDevice", "dspRuntimeBlockPcm", "dspRuntimeUnpackPcmDevice", "dspRuntimeBlockAll", "dspRuntimeBlockAllDevice", "dspRuntimeBlockAllPcm",
    "dspRuntimeSyncState", "dspRuntimeUploadState", "dspRuntimeUploadParams", "dspRuntimeSetOption", "dspRuntimeGetOption",
    "dspRuntimeCoreInfo", "dspRuntimeKernelTime", "dspRuntimeLastError", "dspRuntimeRelease", "dspRuntimeReleaseProgram", "dspRuntimeSelect",
    "dspRuntimeSetShard", "dspRuntimeShardInfo", "dspRuntimeTagOutput", "dspRuntimeTagOutputDevice", "dspRuntimeTagOutputReset",
    "dspRuntimeSetInstances", "dspRuntimeBlockAllInstancesDevice", "dspRuntimeInstanceState",
    # thin HIP ABI (include/avdsp_hip.h)
    "avdsp_hip_device_count", "avdsp_hip_set_device", "avdsp_hip_prog_create", "avdsp_hip_prog_destroy",
    "avdsp_hip_prog_add_plan", "avdsp_hip_prog_add_generic", "avdsp_hip_prog_clear_plans", "avdsp_hip_tpdf_reset", "avdsp_hip_upload_words", "avdsp_hip_download_words", "avdsp_hip_zero_words",
    "avdsp_hip_run_block", "avdsp_hip_run_block_host", "avdsp_hip_submit_block_host", "avdsp_hip_wait_block_host", "avdsp_hip_run_levels", "avdsp_hip_run_levels_host", "avdsp_hip_run_levels_pcm_host", "avdsp_hip_unpack_pcm", "avdsp_hip_run_block_pcm_host", "avdsp_hip_profile_enable", "avdsp_hip_profile_read", "avdsp_hip_prog_set_option", "avdsp_hip_tag_output", "avdsp_hip_tag_column_host",
    "avdsp_hip_synchronize", "avdsp_hip_last_error",
    "avdsp_hip_set_instances", "avdsp_hip_run_levels_instances", "avdsp_hip_download_instance_words", "avdsp_hip_ready_timeouts", "avdsp_hip_ready_clear", "avdsp_hip_chain_instances", "avdsp_hip_prog_get_option", "avdsp_hip_last_error_is_ready_timeout", "avdsp_hip_profile_last_pairs",
    "avdsp_hip_plan_add_strands", "avdsp_hip_plan_strands",
]


class AvdspError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 + gcc via avdsp_amd/csrc/Makefile (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC]
    if force:
        subprocess.check_call(args + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, same as the system one).  If our library were loaded first it would pull the
    system runtime, torch would later add its bundled copy, and whichever initialises second sees
    "No HIP GPUs".  Loading torch's copy first (when torch is installed) makes the dynamic loader
    satisfy our DT_NEEDED libamdhip64.so.7 with it, whatever the import order.  C hosts without
    torch simply get the ROCm runtime through the library's RUNPATH."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib() -> C.CDLL:
    """Load the C-ABI library; raises if it is missing (there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C avdsp_amd/csrc).  The product path has no CPU fallback.")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        vp, i32 = C.c_void_p, C.c_int
        L.dspFindCore.restype = vp; L.dspFindCore.argtypes = [vp, i32]
        L.dspFindCoreBegin.restype = vp; L.dspFindCoreBegin.argtypes = [vp]
        L.dspRuntimeReset.restype = i32; L.dspRuntimeReset.argtypes = [i32, i32, i32]
        L.dspRuntimeInit.restype = i32; L.dspRuntimeInit.argtypes = [vp, i32, i32, i32, i32]
        for n in ("dspRuntime_2", "dspRuntime_3", "dspRuntime_4", "dspRuntime_5", "dspRuntime_6"):
            f = getattr(L, n); f.restype = i32; f.argtypes = [vp, vp, vp]
        for n in ("dspRuntimeBlock_2", "dspRuntimeBlock_3", "dspRuntimeBlock_4", "dspRuntimeBlock_5", "dspRuntimeBlock_6"):
            f = getattr(L, n); f.restype = i32; f.argtypes = [vp, vp, vp, i32, i32, vp, i32, i32, i32]
        L.dspRuntimeBlockDevice.restype = i32
        L.dspRuntimeBlockDevice.argtypes = [i32, vp, vp, vp, i32, i32, vp, i32, i32, i32, vp]
        L.dspRuntimeBlockSubmit.restype = i32
        L.dspRuntimeBlockSubmit.argtypes = [i32, vp, vp, vp, i32, i32, vp, i32, i32, i32]
        L.dspRuntimeBlockWait.restype = i32; L.dspRuntimeBlockWait.argtypes = [i32]
        L.dspRuntimeBlockAll.restype = i32
        L.dspRuntimeBlockAll.argtypes = [i32, vp, vp, i32, i32, vp, i32, i32, i32]
        L.dspRuntimeBlockAllDevice.restype = i32
        L.dspRuntimeBlockAllDevice.argtypes = [i32, vp, vp, i32, i32, vp, i32, i32, i32, vp]
        L.dspRuntimeBlockAllPcm.restype = i32
        L.dspRuntimeBlockAllPcm.argtypes = [i32, vp, i32, vp, i32, i32, vp, i32, i32, i32]
        L.dspRuntimeSetInstances.restype = i32; L.dspRuntimeSetInstances.argtypes = [i32]
        L.dspRuntimeBlockAllInstancesDevice.restype = i32
        L.dspRuntimeBlockAllInstancesDevice.argtypes = [i32, vp, vp, i32, i32, C.c_size_t, vp, i32, i32, C.c_size_t, i32, vp]
        L.dspRuntimeInstanceState.restype = i32; L.dspRuntimeInstanceState.argtypes = [i32, vp]
        L.dspRuntimeBlockPcm.restype = i32
        L.dspRuntimeBlockPcm.argtypes = [i32, vp, vp, i32, vp, i32, i32, vp, i32, i32, i32]
        L.dspRuntimeUnpackPcmDevice.restype = i32
        L.dspRuntimeUnpackPcmDevice.argtypes = [i32, vp, vp, C.c_longlong, vp]
        L.dspRuntimeSyncState.restype = i32; L.dspRuntimeSyncState.argtypes = [vp]
        L.dspRuntimeUploadState.restype = i32; L.dspRuntimeUploadState.argtypes = [vp]
        L.dspRuntimeSetOption.restype = i32; L.dspRuntimeSetOption.argtypes = [C.c_char_p, i32]
        L.dspRuntimeGetOption.restype = i32; L.dspRuntimeGetOption.argtypes = [C.c_char_p]
        L.dspRuntimeCoreInfo.restype = i32
        L.dspRuntimeCoreInfo.argtypes = [i32, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.dspRuntimeStrandInfo.restype = i32
        L.dspRuntimeStrandInfo.argtypes = [i32, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.dspRuntimeTagOutput.restype = i32; L.dspRuntimeTagOutput.argtypes = [vp, i32, i32, i32]
        L.dspRuntimeTagOutputDevice.restype = i32; L.dspRuntimeTagOutputDevice.argtypes = [vp, i32, i32, i32, vp]
        L.dspRuntimeTagOutputReset.restype = i32; L.dspRuntimeTagOutputReset.argtypes = [i32]
        L.dspRuntimeSetShard.restype = i32; L.dspRuntimeSetShard.argtypes = [i32, i32]
        L.dspRuntimeShardInfo.restype = i32
        L.dspRuntimeShardInfo.argtypes = [i32, vp] + [C.POINTER(i32)] * 7
        L.dspRuntimeKernelTime.restype = i32
        L.dspRuntimeKernelTime.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(i32)]
        L.dspRuntimeLastError.restype = C.c_char_p
        L.dspRuntimeRelease.restype = None
        L.dspRuntimeReleaseProgram.restype = i32; L.dspRuntimeReleaseProgram.argtypes = [vp]
        L.dspRuntimeSelect.restype = i32; L.dspRuntimeSelect.argtypes = [vp]
        L.dspQNM.restype = C.c_longlong; L.dspQNM.argtypes = [C.c_double, i32, i32]
        L.dspQM64.restype = C.c_longlong; L.dspQM64.argtypes = [C.c_double, i32]
        L.dspQM32.restype = i32; L.dspQM32.argtypes = [C.c_double, i32]
        L.avdsp_hip_device_count.restype = i32
        L.avdsp_hip_last_error.restype = C.c_char_p
        L.avdsp_hip_synchronize.restype = i32; L.avdsp_hip_synchronize.argtypes = [vp]
        _lib = L
    return _lib


PCM_S32, PCM_S24_3LE, PCM_S16 = 0, 1, 2       # include/avdsp_hip.h AVDSP_PCM_*


def sample_dtype(fmt: int):
    return np.float32 if fmt in (5, 6) else np.int32


class Runtime:
    """One loaded program (the library, like the reference, holds one program per process).

    `buf` is the caller-owned contiguous int32 buffer: program words then the state area."""

    def __init__(self, fmt: int, prog_words: np.ndarray, fs: int = 48000, random: int = 0,
                 dither: int = 31, max_size: int | None = None):
        if fmt not in (2, 3, 4, 5, 6):
            raise ValueError("DSP_FORMAT must be one of 2..6")
        self.L = lib()
        self.fmt = fmt
        n = int(prog_words[1]) + max(int(np.int32(prog_words[2])), 0)
        self.buf = np.zeros(max(n, len(prog_words)) + 64, dtype=np.uint32)
        self.buf[:len(prog_words)] = prog_words
        self.rc = self.L.dspRuntimeInit(self.buf.ctypes.data, n if max_size is None else max_size,
                                        fs, random, dither)
        self.cores = []
        if self.rc >= 0:
            self.rundata = self.buf.ctypes.data + 4 * self.rc
            k = 1
            while True:
                p = self.L.dspFindCore(self.buf.ctypes.data, k)
                if not p:
                    break
                self.cores.append(self.L.dspFindCoreBegin(p))
                k += 1

    # -- helpers ---------------------------------------------------------------------------
    def last_error(self) -> str:
        return self.L.dspRuntimeLastError().decode()

    def _check(self, rc: int) -> int:
        if rc < 0:
            raise AvdspError(rc, self.last_error())
        return rc

    def _select(self):
        """Several programs may be loaded in one process: calls that carry no program pointer address the one selected last."""
        if self.rc >= 0:
            self.L.dspRuntimeSelect(self.buf.ctypes.data)

    @property
    def state(self) -> np.ndarray:
        return self.buf[self.rc:self.rc + int(self.buf[2])]

    def reset(self, fs: int, random: int = 0, dither: int = 31) -> int:
        self._select()
        return self.L.dspRuntimeReset(fs, random, dither)

    def set_option(self, key: str, value: int):
        self._select()
        self._check(self.L.dspRuntimeSetOption(key.encode(), value))

    @staticmethod
    def set_global_option(key: str, value: int):
        """dspRuntimeSetOption without a loaded program (options are process-wide, like the library's state)."""
        if lib().dspRuntimeSetOption(key.encode(), value) < 0:
            raise AvdspError(-1, f"unknown option {key!r}")

    def core_info(self, core_index: int = 0):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self.L.dspRuntimeCoreInfo(self.fmt, self.cores[core_index], C.byref(a), C.byref(b), C.byref(c)))
        return dict(chains=a.value, max_sections=b.value, max_taps=c.value)

    def strand_info(self, core_index: int = 0):
        """dspRuntimeStrandInfo (host-only): does the core's tail run as a strand plan?"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self.L.dspRuntimeStrandInfo(self.fmt, self.cores[core_index], C.byref(a), C.byref(b), C.byref(c)))
        return dict(strands=a.value, ops=b.value, prefix_words=c.value)

    def tag_output(self, out: np.ndarray, column: int):
        """dspRuntimeTagOutput on a host block [frames][out_stride] of int32, in place (linux/avdsp_plugin.c:133-137)."""
        self._select()
        assert out.dtype == np.int32 and out.flags.c_contiguous
        self._check(self.L.dspRuntimeTagOutput(out.ctypes.data, out.shape[1], column, out.shape[0]))

    def set_shard(self, rank: int, world: int):
        """dspRuntimeSetShard: this process runs chains shard_range(total, world, rank) of every chain core."""
        self._select()
        self._check(self.L.dspRuntimeSetShard(rank, world))

    def shard_info(self, core_index: int = 0):
        """dspRuntimeShardInfo: the chains this process runs and the IO numbers they load / store (host-only)."""
        v = [C.c_int() for _ in range(7)]
        self._check(self.L.dspRuntimeShardInfo(self.fmt, self.cores[core_index], *[C.byref(x) for x in v]))
        keys = ("total_chains", "first_chain", "nchains", "in_io_min", "in_io_max", "out_io_min", "out_io_max")
        return {k: x.value for k, x in zip(keys, v)}

    # -- execution -------------------------------------------------------------------------
    def run_frame(self, samples: np.ndarray, core_index: int = 0) -> int:
        """dspRuntime_N: one frame, samples[] indexed by IO number, updated in place."""
        assert samples.dtype == sample_dtype(self.fmt) and samples.flags.c_contiguous
        f = getattr(self.L, f"dspRuntime_{self.fmt}")
        return self._check(f(self.cores[core_index], self.rundata, samples.ctypes.data))

    def run_block(self, x: np.ndarray, out_stride: int, in_io_base: int, out_io_base: int = 0,
                  out: np.ndarray | None = None, block: int | None = None) -> np.ndarray:
        """dspRuntimeBlock_N over host buffers; cores outer, frames inner per block of `block` frames."""
        x = np.ascontiguousarray(x, dtype=sample_dtype(self.fmt))
        nframes, in_stride = x.shape
        if out is None:
            out = np.zeros((nframes, out_stride), dtype=sample_dtype(self.fmt))
        f = getattr(self.L, f"dspRuntimeBlock_{self.fmt}")
        block = block or nframes
        for b0 in range(0, nframes, block):
            b1 = min(b0 + block, nframes)
            for core in self.cores:
                self._check(f(core, self.rundata, x[b0:b1].ctypes.data, in_stride, in_io_base,
                              out[b0:b1].ctypes.data, out_stride, out_io_base, b1 - b0))
        return out

    def submit_block(self, x: np.ndarray, out: np.ndarray, in_io_base: int, out_io_base: int = 0) -> int:
        """dspRuntimeBlockSubmit: the block is queued (copies and kernels of up to four blocks overlap); x and out
        (C-contiguous, the runtime's sample type) belong to the library until wait_blocks lets the block through -- keep
        them alive until then.  With set_option("host_pin", 1) their registrations are kept for reuse and the arrays must
        stay alive until host_pin goes back to 0 (or release()).
        The queue overlaps blocks of SINGLE-core programs (the BASELINE chain programs).  A program with several cores shares
        one output window between them, so core k+1 is only submitted after dspRuntimeBlockWait(0) for core k -- which drains
        every block in flight, not just that core's: such a program runs through the queue synchronously (use run_block_all,
        which arranges the cores on the device, for those).  Returns the number of blocks in flight after the last core's
        submit (for a multi-core program: that core's block alone)."""
        dt = sample_dtype(self.fmt)
        if x.dtype != dt or out.dtype != dt or not x.flags.c_contiguous or not out.flags.c_contiguous:
            raise ValueError("submit_block takes C-contiguous arrays of the runtime's sample type (they are used in place)")
        rc = 0
        for k, core in enumerate(self.cores):
            if k:
                self._check(min(self.L.dspRuntimeBlockWait(0), 0))
            rc = self.L.dspRuntimeBlockSubmit(self.fmt, core, self.rundata, x.ctypes.data, x.shape[1], in_io_base,
                                              out.ctypes.data, out.shape[1], out_io_base, x.shape[0])
            self._check(min(rc, 0))
        return rc

    def wait_blocks(self, max_in_flight: int = 0) -> int:
        """dspRuntimeBlockWait: returns when at most max_in_flight submitted blocks are unfinished."""
        self._select()
        rc = self.L.dspRuntimeBlockWait(max_in_flight)
        self._check(min(rc, 0))
        return rc

    def run_block_all(self, x: np.ndarray, out_stride: int, in_io_base: int, out_io_base: int = 0,
                      out: np.ndarray | None = None, block: int | None = None) -> np.ndarray:
        """dspRuntimeBlockAll: every core of the program per block of `block` frames, in one call each; cores
        that do not meet run at the same time.  Same result as run_block."""
        x = np.ascontiguousarray(x, dtype=sample_dtype(self.fmt))
        nframes, in_stride = x.shape
        if out is None:
            out = np.zeros((nframes, out_stride), dtype=sample_dtype(self.fmt))
        block = block or nframes
        for b0 in range(0, nframes, block):
            b1 = min(b0 + block, nframes)
            self._check(self.L.dspRuntimeBlockAll(self.fmt, self.rundata, x[b0:b1].ctypes.data, in_stride, in_io_base,
                                                  out[b0:b1].ctypes.data, out_stride, out_io_base, b1 - b0))
        return out

    def set_instances(self, n: int):
        """dspRuntimeSetInstances: n copies of the program side by side (instance i: its own state and sample blocks)."""
        self._select()
        self._check(self.L.dspRuntimeSetInstances(n))

    def run_block_all_instances_device(self, d_in_ptr: int, in_stride: int, in_io_base: int, in_inst_words: int,
                                       d_out_ptr: int, out_stride: int, out_io_base: int, out_inst_words: int,
                                       nframes: int, stream: int = 0) -> int:
        """dspRuntimeBlockAllInstancesDevice: every core of every instance over one block each, resident in HBM."""
        return self._check(self.L.dspRuntimeBlockAllInstancesDevice(
            self.fmt, self.rundata, d_in_ptr, in_stride, in_io_base, in_inst_words,
            d_out_ptr, out_stride, out_io_base, out_inst_words, nframes, stream))

    def instance_state(self, inst: int) -> np.ndarray:
        """the data area of instance `inst` (what sync_state() brings back for instance 0)"""
        self._select()
        out = np.zeros(len(self.state), dtype=self.state.dtype)
        self._check(self.L.dspRuntimeInstanceState(inst, out.ctypes.data))
        return out

    def run_block_all_pcm(self, pcm: int, raw: np.ndarray, nframes: int, in_stride: int, out_stride: int,
                          in_io_base: int, out_io_base: int = 0, block: int | None = None) -> np.ndarray:
        """dspRuntimeBlockAllPcm: packed PCM bytes in, every core, S32 out."""
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        width = {PCM_S32: 4, PCM_S24_3LE: 3, PCM_S16: 2}[pcm]
        assert raw.size == nframes * in_stride * width
        out = np.zeros((nframes, out_stride), dtype=np.int32)
        block = block or nframes
        for b0 in range(0, nframes, block):
            b1 = min(b0 + block, nframes)
            src = raw[b0 * in_stride * width:b1 * in_stride * width]
            self._check(self.L.dspRuntimeBlockAllPcm(self.fmt, self.rundata, pcm, src.ctypes.data, in_stride, in_io_base,
                                                     out[b0:b1].ctypes.data, out_stride, out_io_base, b1 - b0))
        return out

    def get_option(self, key: str) -> int:
        self._select()
        return self.L.dspRuntimeGetOption(key.encode())

    def run_block_pcm(self, pcm: int, raw: np.ndarray, nframes: int, in_stride: int, out_stride: int,
                      in_io_base: int, out_io_base: int = 0, block: int | None = None) -> np.ndarray:
        """dspRuntimeBlockPcm: `raw` = packed PCM bytes (uint8) of nframes x in_stride samples."""
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        width = {PCM_S32: 4, PCM_S24_3LE: 3, PCM_S16: 2}[pcm]
        assert raw.size == nframes * in_stride * width
        out = np.zeros((nframes, out_stride), dtype=np.int32)
        block = block or nframes
        for b0 in range(0, nframes, block):
            b1 = min(b0 + block, nframes)
            src = raw[b0 * in_stride * width:b1 * in_stride * width]
            for core in self.cores:
                self._check(self.L.dspRuntimeBlockPcm(self.fmt, core, self.rundata, pcm, src.ctypes.data, in_stride,
                                                      in_io_base, out[b0:b1].ctypes.data, out_stride, out_io_base, b1 - b0))
        return out

    def run_block_device(self, d_in_ptr: int, in_stride: int, in_io_base: int, d_out_ptr: int,
                         out_stride: int, out_io_base: int, nframes: int, stream: int = 0,
                         core_index: int = 0) -> int:
        """dspRuntimeBlockDevice: in/out already in HBM (raw device pointers, e.g. tensor.data_ptr())."""
        return self._check(self.L.dspRuntimeBlockDevice(
            self.fmt, self.cores[core_index], self.rundata, d_in_ptr, in_stride, in_io_base,
            d_out_ptr, out_stride, out_io_base, nframes, stream))

    def kernel_time(self, kind: int):
        """(total_ms, launches) of the kernels of `kind` (0 biquad, 1 FIR, 2 pass) since the last read;
        needs set_option("profile", 1)."""
        self._select()
        ms, n = C.c_double(), C.c_int()
        self._check(self.L.dspRuntimeKernelTime(kind, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def sync_state(self) -> np.ndarray:
        self._check(self.L.dspRuntimeSyncState(self.rundata))
        return self.state

    def upload_state(self):
        self._check(self.L.dspRuntimeUploadState(self.rundata))

    def upload_params(self):
        """After editing parameter words in self.buf in place."""
        self._select()
        self._check(self.L.dspRuntimeUploadParams())

    def release(self):
        """Frees this program's device memory (dspRuntimeReleaseProgram); other loaded programs stay."""
        if self.rc >= 0:
            self.L.dspRuntimeReleaseProgram(self.buf.ctypes.data)

    def __del__(self):
        try:                                                  # the buffer goes with this object: so does the program's context
            self.release()
        except Exception:
            pass
