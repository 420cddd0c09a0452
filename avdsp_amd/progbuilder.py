"""Builder for AVDSP encoded programs (.bin).

Restates the word layout the reference encoder emits (module_avdsp/encoder/dsp_encoder.c) for the
subset of opcodes on the hot path, so that synthetic many-channel programs can be produced without
the reference being present (the GPU box has no /root/reference).  For biquad-only programs the
output is byte-identical to the reference encoder library driven through its public API
(checked by tests/test_progbuilder.py against fixtures made by tests/golden/make_goldens.py).

Layout facts restated here (reference file:line):
  * head word = (opcode << 16) | skip                          dsp_encoder.c:107-109
  * header, 12 words, version 0x102                            dsp_encoder.c:12,333-380,509-548
  * CORE = [op][usedInputsCore][usedOutputsCore]               dsp_encoder.c:624-632,454-462
  * LOAD_GAIN fixed = [op][IO][3][gain]                        dsp_encoder.c:765-780
  * biquad bank inside a PARAM, at an ODD word index:
      [(50<<16)|nSec][bypass=1] then per section
      [(type<<16)|freq][Q f32][gain f32] then per sample rate [b0 b1 b2 a1-1 a2] + 1 pad word
                                                               dsp_encoder.c:1225-1290
  * BIQUADS = [op][dataOff][bankOff rel. to op], 6 state words per section, 8-byte aligned
                                                               dsp_encoder.c:1212-1223
  * FIR impulses inside a PARAM: [(51<<16)|nF] at an odd index, then per rate an odd-indexed
      length word followed by `length` float taps            dsp_encoder.c:1292-1298,1347-1372
  * FIR = [op][impOff per rate, rel. to op][dataOff]; the runtime (dsp_runtime.c:928-969) expects
      impOff to address the LENGTH word.  The reference's own dsp_FIR() addresses the section header
      instead (dsp_encoder.c:1311-1313, an encoder bug), so this builder follows the runtime.
  * END_OF_CODE = [0] (+1 pad word to make the length even)    dsp_encoder.c:509-516
  * header.checkSum = sum of head words                        runtime/dsp_header.h:234-251
Peaking-EQ coefficients follow encoder/dsp_filters.c:94-102,135-143 (a1 is stored minus 1.0).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# opcode numbers (runtime/dsp_header.h:40-132)
OP_END, OP_HEADER, OP_NOP, OP_CORE, OP_PARAM = 0, 1, 2, 3, 4
OP_SWAPXY, OP_COPYXY = 11, 12
OP_LOAD, OP_LOAD_GAIN, OP_STORE = 34, 35, 37
OP_GAIN, OP_SAT0DB = 41, 42
OP_BIQUADS, OP_FIR = 50, 51

FREQ_TABLE = (8000, 16000, 24000, 32000, 44100, 48000, 88200, 96000,
              176400, 192000, 352800, 384000, 705600, 768000)
F44100, F48000, F96000, F192000 = 4, 5, 7, 9
FPEAK = 74                      # encoder/dsp_filters.h:14-25 (enum position)
ENCODER_VERSION = 0x102         # dsp_encoder.c:12
MANT = 28                       # runtime/dsp_header.h:258-267


def _f32_bits(x) -> int:
    return int(np.float32(x).view(np.uint32))


def _qm32(x: float, m: int = MANT) -> int:
    """DSP_QM32 (runtime/dsp_header.h:276-284): saturating double -> Q(32-m).m, truncating."""
    lim = float(1 << (32 - m - 1))
    if x >= lim:
        return 0x7FFFFFFF
    if -x > lim:
        return 0x80000000
    return int(x * float(1 << m)) & 0xFFFFFFFF


def peaking_coefs(fs: float, f0: float, q: float, gain_f32: float):
    """encoder/dsp_filters.c:94-102,135-143; returns (b0, b1, b2, a1, a2) as Python doubles.

    a1/a2 already carry the sign convention of the reference (recurrence is purely additive);
    the stored value of a1 is a1 - 1.0 (dsp_encoder.c:1280,1286)."""
    w0 = math.pi * 2.0 * f0 / fs
    cw0 = math.cos(w0)
    sw0 = math.sin(w0)
    alpha = sw0 / 2.0 / q if q != 0.0 else 1.0
    a = math.sqrt(gain_f32)
    a0 = 1.0 + alpha / a
    a1 = 2.0 * cw0 / a0
    a2 = -(1.0 - alpha / a) / a0
    b0 = (1.0 + alpha * a) / a0
    b1 = -2.0 * cw0 / a0
    b2 = (1.0 - alpha * a) / a0
    return b0, b1, b2, a1, a2


@dataclass
class Section:
    """One biquad cell: user parameters + one (b0,b1,b2,a1,a2) tuple per encoded sample rate."""
    ftype: int
    freq: float
    q: float
    gain: float
    coefs: list          # [(b0,b1,b2,a1,a2)] * nF, a1 NOT yet reduced by 1.0


@dataclass
class ProgramWriter:
    """Emits program words in the order and with the padding rules of the reference encoder."""
    fmt: int                      # DSP_FORMAT of the runtime that will load it (2..6)
    fmin: int = F48000
    fmax: int = F48000
    capacity: int = 1 << 16
    words: np.ndarray = field(init=False)
    idx: int = field(init=False, default=0)
    data_counter: int = field(init=False, default=0)
    max_opcode: int = field(init=False, default=0)
    used_in: int = field(init=False, default=0)
    used_out: int = field(init=False, default=0)
    core_in: int = field(init=False, default=0)
    core_out: int = field(init=False, default=0)
    last_core: int = field(init=False, default=0)
    open_param: int = field(init=False, default=-1)

    def __post_init__(self):
        self.words = np.zeros(self.capacity, dtype=np.uint32)
        self.int_mode = self.fmt < 3                       # dsp_encoder.c:327-330
        self.nf = self.fmax - self.fmin + 1
        self.idx = 12                                       # header filled in by end_of_code()

    # -- primitives -------------------------------------------------------------------------
    def _w(self, v: int) -> int:
        i = self.idx
        self.words[i] = v & 0xFFFFFFFF
        self.idx += 1
        return i

    def _wf(self, x: float) -> int:
        return self._w(_f32_bits(x))

    def _param_value(self, x: float) -> int:
        """addGainCodeQNM (dsp_encoder.c:608-618): Q28 when int-encoded, float otherwise."""
        return self._w(_qm32(x)) if self.int_mode else self._wf(x)

    def _close_param(self):
        if self.open_param >= 0:
            p = self.open_param
            self.words[p] = (OP_PARAM << 16) | ((self.idx - p) & 0xFFFF)
            self.open_param = -1

    def _head(self, op: int, skip: int) -> int:
        self._close_param()
        self.max_opcode = max(self.max_opcode, op)
        return self._w((op << 16) | skip)

    def _data_aligned8(self, size: int) -> int:             # dsp_encoder.c:141-144
        if self.data_counter & 1:
            self.data_counter += 1
        off = self.data_counter
        self.data_counter += size
        return off

    def _flush_core_io(self):                               # dsp_encoder.c:454-462
        if self.last_core:
            self.words[self.last_core + 1] = self.core_in
            self.words[self.last_core + 2] = self.core_out
            self.last_core = 0

    # -- opcodes ----------------------------------------------------------------------------
    def core(self):
        self._flush_core_io()
        self.core_in = self.core_out = 0
        self.last_core = self._head(OP_CORE, 3)
        self.idx += 2

    def param(self) -> int:
        p = self._head(OP_PARAM, 0)
        self.open_param = p
        return p

    def biquad_bank(self, sections: list, bypass: int = 1) -> int:
        """dspBiquad_Sections + dsp_Filter2ndOrder... ; returns the bank's word index."""
        if (self.idx & 1) == 0:
            self._w(0)                                       # paramMisAligned8
        pos = self._w((OP_BIQUADS << 16) | (len(sections) & 0xFFFF))
        self._w(bypass)
        for s in sections:
            self._w((s.ftype << 16) | (int(s.freq) & 0xFFFF))
            self._wf(s.q)
            self._wf(s.gain)
            for (b0, b1, b2, a1, a2) in s.coefs:
                if self.idx & 1:
                    self._w(0)                               # paramAligned8
                for v in (b0, b1, b2, a1 - 1.0, a2):
                    self._w(_qm32(v)) if self.int_mode else self._wf(v)
        return pos

    def fir_impulses(self, taps_per_rate: list) -> list:
        """Returns the word index of each rate's LENGTH word (what DSP_FIR must point at)."""
        if (self.idx & 1) == 0:
            self._w(0)
        self._w((OP_FIR << 16) | self.nf)
        where = []
        for taps in taps_per_rate:
            if (self.idx & 1) == 0:
                self._w(0)
            taps = np.asarray(taps, dtype=np.float32)
            where.append(self._w(len(taps)))
            n = len(taps)
            self.words[self.idx:self.idx + n] = taps.view(np.uint32)
            self.idx += n
        return where

    def _mark_in(self, io: int):
        if io < 32:
            self.used_in |= 1 << io
            self.core_in |= 1 << io

    def load(self, io: int):
        self._head(OP_LOAD, 2)
        self._mark_in(io)
        self._w(io)

    def load_gain_fixed(self, io: int, gain: float):
        self._head(OP_LOAD_GAIN, 4)
        self._mark_in(io)
        self._w(io)
        self._w(3)
        self._param_value(gain)

    def gain_fixed(self, gain: float):
        self._head(OP_GAIN, 3)
        self._w(2)
        self._param_value(gain)

    def biquads(self, bank: int, nsections: int) -> int:
        base = self._head(OP_BIQUADS, 3)
        off = self._data_aligned8(6 * nsections)
        self._w(off)
        self._w(bank - base)
        return off

    def fir(self, length_words: list, max_len: int) -> int:
        base = self._head(OP_FIR, 2 + self.nf)
        for w in length_words:
            self._w((w - base) if w else 0)
        off = self._data_aligned8(max_len)
        self._w(off)
        return off

    def sat0db(self):
        self._head(OP_SAT0DB, 1)

    def copyxy(self):
        self._head(OP_COPYXY, 1)

    def swapxy(self):
        self._head(OP_SWAPXY, 1)

    def store(self, io: int):
        self._head(OP_STORE, 2)
        self._w(io)
        if io < 32:
            self.used_out |= 1 << io
            self.core_out |= 1 << io

    def end_of_code(self) -> np.ndarray:
        self._flush_core_io()
        self._head(OP_END, 0)
        if self.idx & 1:
            self._w(0)
        n = self.idx
        w = self.words
        w[0] = (OP_HEADER << 16) | 12
        w[1] = n
        w[2] = self.data_counter
        w[5] = ENCODER_VERSION
        w[6] = ((MANT if self.int_mode else 0) & 0xFFFF) | (self.max_opcode << 16)
        w[7] = self.fmin
        w[8] = self.fmax
        w[9] = self.used_in & 0xFFFFFFFF
        w[10] = self.used_out & 0xFFFFFFFF
        w[11] = 0
        out = w[:n].copy()
        s, cores = checksum(out)
        out[3] = s
        out[4] = cores
        return out


def checksum(words: np.ndarray):
    """runtime/dsp_header.h:234-251: sum of head words, number of cores."""
    pos, total, cores = 0, 0, 0
    n = len(words)
    while pos < n:
        w = int(words[pos])
        skip = w & 0xFFFF
        if skip == 0:
            break
        if (w >> 16) == OP_CORE:
            cores += 1
        total = (total + w) & 0xFFFFFFFF
        pos += skip
    return total, max(cores, 1)


def lcg_taps(channel: int, ntaps: int) -> np.ndarray:
    """s = s*1664525+1013904223 (seed 7+c) -> ((int32)s / 2^31) * 4 / T as float."""
    s = np.empty(ntaps, dtype=np.uint32)
    v = (7 + channel) & 0xFFFFFFFF
    for i in range(ntaps):
        v = (v * 1664525 + 1013904223) & 0xFFFFFFFF
        s[i] = v
    x = s.view(np.int32).astype(np.float64) / 2147483648.0 * 4.0 / ntaps
    return x.astype(np.float32)


def lcg_taps_all(channels: int, ntaps: int, channel_base: int = 0) -> np.ndarray:
    """Vectorised lcg_taps for channels channel_base .. channel_base+channels-1 -> float32 [channels, ntaps]."""
    v = (7 + channel_base + np.arange(channels, dtype=np.uint64)) & 0xFFFFFFFF
    out = np.empty((channels, ntaps), dtype=np.uint32)
    for i in range(ntaps):
        v = (v * 1664525 + 1013904223) & 0xFFFFFFFF
        out[:, i] = v
    x = out.view(np.int32).astype(np.float64) / 2147483648.0 * 4.0 / ntaps
    return x.astype(np.float32)


def lcg_input(nframes: int, nch: int, float_samples: bool, seed: int = 12345) -> np.ndarray:
    """frame-interleaved [B][C]; x = (int32)s >> 3, or that / 2^31 as float."""
    n = nframes * nch
    # closed-form stepping of the LCG, vectorised: s_k = a^k s_0 + c (a^k - 1)/(a - 1)  (mod 2^32)
    out = np.empty(n, dtype=np.uint32)
    v = seed & 0xFFFFFFFF
    chunk = 1 << 16
    a, c = 1664525, 1013904223
    # per-chunk: generate first `chunk` multipliers once
    mul = np.empty(chunk, dtype=np.uint64)
    add = np.empty(chunk, dtype=np.uint64)
    m, d = 1, 0
    for i in range(chunk):
        m = (m * a) & 0xFFFFFFFF
        d = (d * a + c) & 0xFFFFFFFF
        mul[i] = m
        add[i] = d
    pos = 0
    while pos < n:
        k = min(chunk, n - pos)
        blk = (mul[:k] * np.uint64(v) + add[:k]) & np.uint64(0xFFFFFFFF)
        out[pos:pos + k] = blk.astype(np.uint32)
        v = int(out[pos + k - 1])
        pos += k
    x = out.view(np.int32) >> 3
    if float_samples:
        return (x.astype(np.float64) / 2147483648.0).astype(np.float32).reshape(nframes, nch)
    return x.astype(np.int32).reshape(nframes, nch)


def synth_sections(channel: int, nsections: int, fmin: int, fmax: int, _cache={}) -> list:
    """FPEAK, f0 = 100+37b+3(c mod 97), Q = 0.7+0.05(b mod 5), gain 1.2/0.8."""
    out = []
    for b in range(nsections):
        key = (b, channel % 97, fmin, fmax)
        sec = _cache.get(key)
        if sec is None:
            f0 = 100.0 + 37.0 * b + 3.0 * (channel % 97)
            q = 0.7 + 0.05 * (b % 5)
            g = float(np.float32(1.2 if (b & 1) else 0.8))
            coefs = [peaking_coefs(float(FREQ_TABLE[f]), f0, q, g) for f in range(fmin, fmax + 1)]
            sec = Section(FPEAK, f0, q, g, coefs)
            _cache[key] = sec
        out.append(sec)
    return out


def synth_program(fmt: int, channels: int, nsections: int, ntaps: int = 0,
                  fmin: int = F48000, fmax: int = F48000, gain: float = 1.0,
                  shared_taps: bool = False, taps: np.ndarray | None = None,
                  channel_base: int = 0) -> np.ndarray:
    """The synthetic workload of SURVEY.md 8(d) / BASELINE.json:

        channel c:  PARAM{bank_c [, impulse_c]}  LOAD_GAIN(IO=C+c, gain)  BIQUADS(bank_c)
                    [FIR(impulse_c)]  SAT0DB  STORE(IO=c)       one CORE, inputs at IO C..2C-1.

    Returns the program words (uint32, length = header.totalLength); the caller appends
    header.dataSize words of state space.  `taps` overrides the LCG impulses ([C, T] float32).
    `channel_base` makes this the shard [channel_base, channel_base+channels) of a larger program:
    filters and impulses are those of the global channel numbers, IO numbers stay local."""
    nf = fmax - fmin + 1
    per_ch = 16 + nsections * (2 + 6 * nf) + 8 + nf * (ntaps + 3) + 8
    pw = ProgramWriter(fmt, fmin, fmax, capacity=32 + channels * per_ch)
    if ntaps and taps is None:
        taps = lcg_taps_all(1 if shared_taps else channels, ntaps, channel_base)
    pw.core()
    for c in range(channels):
        pw.param()
        bank = pw.biquad_bank(synth_sections(channel_base + c, nsections, fmin, fmax)) if nsections else None
        imp = None
        if ntaps:
            t = taps[0 if shared_taps else c]
            imp = pw.fir_impulses([t] * nf)
        pw.load_gain_fixed(channels + c, gain)
        if bank is not None:
            pw.biquads(bank, nsections)
        if imp is not None:
            pw.fir(imp, ntaps)
        pw.sat0db()
        pw.store(c)
    return pw.end_of_code()


def with_data_area(prog: np.ndarray) -> np.ndarray:
    """Program words followed by header.dataSize zeroed state words (what dspRuntimeInit expects)."""
    buf = np.zeros(int(prog[1]) + int(prog[2]), dtype=np.uint32)
    buf[:len(prog)] = prog
    return buf
