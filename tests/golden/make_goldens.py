#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the COMPILED REFERENCE.

Runs only in the build container (needs /root/reference and oracle/_ref built by
oracle/build_ref.sh).  The fixtures are data -- program words, seeded inputs, the reference
runtime's outputs/state, return codes -- never reference source.  Re-run after changing
progbuilder.py's layout:  python tests/golden/make_goldens.py

Every case is executed by oracle/_ref/ref_driver (the reference runtime, cores outer / frames
inner as in linux/avdsp_plugin.c:95-142) or oracle/_ref/refk_N.so (the reference's kernels from
its unmodified headers).  Large outputs are stored as SHA-256 plus the first/last 16 frames.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from avdsp_amd import progbuilder as pb          # noqa: E402
from oracle import pyoracle as po                # noqa: E402
from tests.golden_recipes import make_program, make_input   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("AVDSP_REFERENCE", "/root/reference") + "/module_avdsp"
REFBIN = po.REF_DIR


FUZZ_SEEDS = 12
ENC_SWEEP_VARIANTS = [(2, 4, 9), (6, 4, 7), (6, 0, 13), (2, 5, 5)]     # (encoding, freqMin index, freqMax index)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def synth(fmt, channels, sections, taps=0, fmin=pb.F48000, fmax=pb.F48000, gain=1.0):
    return dict(kind="synth", fmt=fmt, channels=channels, sections=sections, taps=taps,
                fmin=fmin, fmax=fmax, gain=gain)


def lcg(frames, channels, seed=12345):
    return dict(kind="lcg", frames=frames, channels=channels, seed=seed)


def run_case(name, fmt, prog_recipe, in_recipe, out_stride, in_base, out_base=0, fs=48000, random=0,
             dither=31, block=None, scratch=None, full=True, manifest=None):
    prog = make_program(prog_recipe)
    x = make_input(in_recipe, fmt)
    rc, out, buf = po.run_reference(fmt, prog, x, out_stride, in_base, out_base, fs=fs,
                                    random=random, dither=dither, block=block,
                                    scratch_len=scratch, want_state=True)
    assert rc >= 0, (name, rc)
    state = buf[rc:rc + int(prog[2])]
    entry = dict(name=name, program=prog_recipe, input=in_recipe, fmt=fmt, fs=fs, random=random, dither=dither, block=block or len(x),
                 scratch=scratch, out_stride=out_stride, in_base=in_base, out_base=out_base,
                 init_rc=rc, nframes=int(x.shape[0]), in_stride=int(x.shape[1]),
                 out_sha=sha(out), state_sha=sha(state), prog_sha=sha(prog), in_sha=sha(x), full=full)
    arrays = dict(head=out[:16], tail=out[-16:])
    if full:
        arrays.update(out=out, state=state)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    manifest.append(entry)
    print(f"  {name}: rc={rc} out={out.shape} {'full' if full else 'hash'}")


def main():
    if not po.have_ref():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` in the build container first")
    os.makedirs(OUT, exist_ok=True)
    manifest = []

    # ---- G1/G2: programs committed by the reference (osx/*.bin), copied as data fixtures ----
    print("committed programs")
    for fname, fmt in [("crossoverLV6.bin", 2), ("dacdiy1.bin", 2), ("dsptest1.bin", 3),
                       ("dacfabriceo.bin", 2), ("mydspcode.bin", 2)]:
        shutil.copyfile(os.path.join(REF, "osx", fname), os.path.join(OUT, fname))
        frames = 2000 if fname == "crossoverLV6.bin" else 400
        for block in (1, 64):
            run_case(f"{fname[:-4]}_fs48000_b{block}", fmt, dict(kind="file", name=fname),
                     lcg(frames, 16, seed=999), 32, 8, 0, fs=48000, random=12345, dither=24,
                     block=block, scratch=40,
                     full=(block == 1 and fname != "crossoverLV6.bin") or fname == "mydspcode.bin",
                     manifest=manifest)
    run_case("crossoverLV6_fs96000_b1", 2, dict(kind="file", name="crossoverLV6.bin"),
             lcg(300, 16, seed=4242), 32, 8, 0, fs=96000, random=7, dither=24, block=1,
             scratch=40, manifest=manifest)

    # ---- opcode tour: every opcode the committed programs do not reach, encoded by the reference
    #      encoder (oracle/ref_encode_ops.c) and run by the reference runtimes in all five formats.
    #      (DSP_DCBLOCK in the float-accumulator formats 3 and 5: the reference's own -Ofast build
    #      evaluates acc + (x - x1) as (acc + x) - x1; restatements follow the binary, see oracle_interp.inc.)
    print("opcode tour")
    for enc_fmt, fname, fmts in ((2, "tour_int.bin", [2]), (6, "tour_float.bin", [3, 4, 5, 6])):
        subprocess.check_call([os.path.join(REFBIN, "ref_encode_ops"), str(enc_fmt), os.path.join(OUT, fname)],
                              stdout=subprocess.DEVNULL)
        for fmt in fmts:
            for fs, block, full in ((48000, 64, True), (48000, 1, False), (96000, 1200, False)):
                run_case(f"{fname[:-4]}_f{fmt}_fs{fs}_b{block}", fmt, dict(kind="file", name=fname),
                         lcg(1200, 16, seed=77), 32, 32, 0, fs=fs, random=12345, dither=24, block=block,
                         scratch=48, full=full, manifest=manifest)

    # ---- random well-formed programs (tests/fuzz_programs.py), all five models: several cores, X/Y
    #      arithmetic, filters, delay lines, meters, dither ... in random order.  Before the fixtures are
    #      written the same programs are encoded with the REFERENCE encoder as well and must come out
    #      byte-identical (the generator itself drives this repository's encoder).
    print("random programs")
    from tests.fuzz_programs import random_program, N_IN, IN_BASE, N_OUT
    # (in a child process: the reference library is built -Ofast, and merely loading it switches the loading
    # thread to flush-to-zero arithmetic, which would corrupt the subnormal test inputs made further down)
    check = ("import sys; sys.path.insert(0, %r); from tests.fuzz_programs import random_program\n"
             "for seed in range(%d):\n"
             "    for fmt in (2, 6):\n"
             "        a, b = random_program(seed, fmt), random_program(seed, fmt, %r)\n"
             "        assert len(a) == len(b) and (a == b).all(), ('encoder mismatch', seed, fmt)\n"
             % (ROOT, FUZZ_SEEDS, os.path.join(REFBIN, "libavdspencoder.so")))
    subprocess.check_call([sys.executable, "-c", check])
    for seed in range(FUZZ_SEEDS):
        for fmt in (2, 3, 4, 5, 6):
            run_case(f"fuzz_s{seed}_f{fmt}", fmt, dict(kind="fuzz", seed=seed, fmt=fmt), lcg(700, N_IN, seed=seed + 5),
                     N_OUT, IN_BASE, 0, fs=[48000, 48000, 96000][seed % 3], random=seed * 7 + 1, dither=24,
                     block=[1, 64, 700][seed % 3], scratch=48, full=seed < 2, manifest=manifest)

    # ---- encoder workout (oracle/enc_sweep.c) through the reference encoder: byte fixtures for
    #      avdsp_amd/csrc/avdsp_encoder.c (tests/test_encoder.py) ----
    print("encoder sweep")
    for fmt, fmin, fmax in ENC_SWEEP_VARIANTS:
        subprocess.check_call([os.path.join(REFBIN, "enc_sweep"), str(fmt), str(fmin), str(fmax),
                               os.path.join(OUT, f"enc_sweep_{fmt}_{fmin}_{fmax}.bin")], stdout=subprocess.DEVNULL)

    # ---- the subnormal range: the reference is built -Ofast and runs with MXCSR.FTZ/DAZ.  An impulse
    #      followed by a long silence lets every section decay through the smallest normal numbers into
    #      flushed, signed zeros; "denormals" feeds subnormal samples directly. ----
    print("decay into silence / subnormal inputs")
    for fmt in (2, 3, 4, 5, 6):
        enc_fmt = 2 if fmt == 2 else 6
        run_case(f"decay_c2_s3_f{fmt}", fmt, synth(enc_fmt, 2, 3), dict(kind="impulse", frames=40000, channels=2, value_f=0.5, value_i=1 << 30),
                 2, 2, block=4000, full=False, manifest=manifest)
        run_case(f"denormals_c2_s2_f{fmt}", fmt, synth(enc_fmt, 2, 2), dict(kind="denormals", frames=96, channels=2),
                 2, 2, block=96, manifest=manifest)
        if fmt != 2:
            run_case(f"decay_fir_c2_s2_t33_f{fmt}", fmt, synth(6, 2, 2, 33), dict(kind="impulse", frames=3000, channels=2, value_f=-0.5, value_i=-(1 << 30)),
                     2, 2, block=1000, manifest=manifest)
            run_case(f"denormals_fir_c2_s1_t9_f{fmt}", fmt, synth(6, 2, 1, 9), dict(kind="denormals", frames=96, channels=2),
                     2, 2, block=96, manifest=manifest)

    # ---- reference-encoder byte identity for progbuilder.py ----
    print("reference encoder programs")
    enc = []
    with tempfile.TemporaryDirectory() as d:
        for (fmt, c, s, fmin, fmax) in [(2, 8, 8, 5, 5), (6, 8, 8, 5, 5), (4, 3, 5, 4, 7),
                                        (2, 100, 16, 4, 9), (6, 64, 16, 5, 5), (2, 64, 16, 5, 5)]:
            p = os.path.join(d, "e.bin")
            subprocess.check_call([os.path.join(REFBIN, "ref_encode"), str(fmt), str(c), str(s),
                                   str(fmin), str(fmax), p], stdout=subprocess.DEVNULL)
            w = np.fromfile(p, dtype=np.uint32)
            enc.append(dict(fmt=fmt, channels=c, sections=s, fmin=fmin, fmax=fmax, words=len(w), sha=sha(w)))
            if c <= 8:
                np.save(os.path.join(OUT, f"refenc_f{fmt}_c{c}_s{s}_{fmin}_{fmax}.npy"), w)
    # reproducibility of the three committed .bin through the reference's own dspcreate
    repro = []
    with tempfile.TemporaryDirectory() as d:
        for so, args, target in [
            ("crossoverLV6.so", "-dspformat 2 -fsmax 96000 -fx 800", "crossoverLV6.bin"),
            ("oktodac_diy.so", "-dspformat 2 -fsmax 192000 -prog 1 -dither 24", "dacdiy1.bin"),
            ("testfunction.so", "-dspformat 3 -fsmax 96000 -test1 -dither 26", "dsptest1.bin")]:
            p = os.path.join(d, target)
            subprocess.run([os.path.join(REFBIN, "dspcreate"), "-dspprog", os.path.join(REFBIN, so),
                            "-binfile", p] + args.split(), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, check=True)
            same = open(p, "rb").read() == open(os.path.join(REF, "osx", target), "rb").read()
            repro.append(dict(target=target, identical=bool(same)))
            assert same, target

    # ---- G3: BASELINE config 2 shape, every arithmetic model ----
    print("synthetic biquad cascades")
    for fmt in (2, 3, 4, 5, 6):
        run_case(f"bq_c8_s8_b256_f{fmt}", fmt, synth(fmt, 8, 8), lcg(256, 8), 8, 8, manifest=manifest)
    for fmt in (2, 4, 6):
        run_case(f"bq_c8_s8_fullscale_f{fmt}", fmt, synth(fmt, 8, 8, gain=4.0),
                 dict(kind="fullscale", frames=256, channels=8), 8, 8, manifest=manifest)
        run_case(f"bq_c8_s8_impulse_f{fmt}", fmt, synth(fmt, 8, 8),
                 dict(kind="impulse", frames=256, channels=8, value_f=0.5, value_i=0x40000000),
                 8, 8, manifest=manifest)
    # multi-rate bank, a middle rate selected
    for fmt in (2, 6):
        run_case(f"bq_c5_s3_rates_f{fmt}", fmt, synth(fmt, 5, 3, 0, 4, 7), lcg(200, 5), 5, 5,
                 fs=88200, manifest=manifest)
    # ---- G4: 64 ch x 16 sections x 1024 frames ----
    for fmt in (2, 6):
        run_case(f"bq_c64_s16_b1024_f{fmt}", fmt, synth(fmt, 64, 16), lcg(1024, 64), 64, 64,
                 full=False, manifest=manifest)

    # ---- G5: FIR, float models only (int FIR is undefined behaviour in the reference) ----
    print("FIR")
    for fmt in (4, 6):
        for taps, frames, full in [(7, 128, True), (255, 600, True), (4096, 4608, False)]:
            run_case(f"fir_c4_t{taps}_noise_f{fmt}", fmt, synth(fmt, 4, 0, taps), lcg(frames, 4), 4, 4,
                     full=full, manifest=manifest)
        run_case(f"fir_c4_t255_impulse_f{fmt}", fmt, synth(fmt, 4, 0, 255),
                 dict(kind="impulse", frames=300, channels=4, value_f=0.25, value_i=0x20000000),
                 4, 4, manifest=manifest)
    # ---- G6: mixed chain ----
    run_case("mixed_c16_s8_t2048_f6", 6, synth(6, 16, 8, 2048), lcg(2304, 16), 16, 16, full=False,
             manifest=manifest)
    run_case("mixed_c4_s2_t255_f6", 6, synth(6, 4, 2, 255), lcg(600, 4), 4, 4, manifest=manifest)
    run_case("mixed_c4_s2_t255_f4", 4, synth(4, 4, 2, 255), lcg(600, 4), 4, 4, manifest=manifest)

    # ---- G7: return codes of dspRuntimeInit / dspRuntimeReset ----
    print("negative cases")
    neg = []
    good = pb.synth_program(2, 2, 2)
    x = pb.lcg_input(4, 2, False)

    def rc_of(prog, fmt=2, fs=48000, max_size=0):
        return po.run_reference(fmt, prog, x, 2, 2, fs=fs, max_size=max_size)[0]

    bad = good.copy(); bad[3] ^= 1
    neg.append(dict(case="bad_checksum", rc=rc_of(bad)))
    neg.append(dict(case="unsupported_fs", fs=12345, rc=rc_of(good, fs=12345)))
    neg.append(dict(case="fs_out_of_range", fs=96000, rc=rc_of(good, fs=96000)))
    neg.append(dict(case="buffer_too_small", max_size=int(good[1]) + int(good[2]) - 1,
                    rc=rc_of(good, max_size=int(good[1]) + int(good[2]) - 1)))
    bad = good.copy(); bad[0] = (2 << 16) | 12
    neg.append(dict(case="no_header", rc=rc_of(bad)))
    bad = good.copy(); bad[6] = (62 << 16) | (int(bad[6]) & 0xFFFF)
    neg.append(dict(case="opcode_too_new", rc=rc_of(bad)))
    neg.append(dict(case="ok", rc=rc_of(good)))
    for n in neg:
        print("  ", n)

    # ---- kernel-level vectors from the reference's unmodified headers ----
    print("kernel vectors")
    k6 = C.CDLL(os.path.join(REFBIN, "refk_6.so"))
    k2 = C.CDLL(os.path.join(REFBIN, "refk_2.so"))
    f32, f64, i32, i64 = C.c_float, C.c_double, C.c_int, C.c_longlong
    k6.refk_mul_float_double.restype = f64; k6.refk_mul_float_double.argtypes = [f32, f32]
    k6.refk_mul_float_float.restype = f32; k6.refk_mul_float_float.argtypes = [f32, f32]
    k6.refk_int_to_float_scaled.restype = f32; k6.refk_int_to_float_scaled.argtypes = [i32, i32]
    k6.refk_int_to_double_scaled.restype = f64; k6.refk_int_to_double_scaled.argtypes = [i32, i32]
    k6.refk_s31_from_double.restype = i32; k6.refk_s31_from_double.argtypes = [f64]
    k6.refk_saturate_double.restype = f64; k6.refk_saturate_double.argtypes = [f64]
    k6.refk_truncate_double.restype = f64; k6.refk_truncate_double.argtypes = [f64, i32]
    k2.refk_saturate64_031.restype = i64; k2.refk_saturate64_031.argtypes = [i64, i32]
    rng = np.random.default_rng(20241220)
    n = 4000
    a = (rng.standard_normal(n) * np.exp2(rng.integers(-30, 8, n))).astype(np.float32)
    b = (rng.standard_normal(n) * np.exp2(rng.integers(-30, 8, n))).astype(np.float32)
    a[:8] = [0.0, -0.0, 1e-40, -1e-40, 1.0, -1.0, 3.0e38, 1.1754944e-38]
    b[:8] = [1.0, 2.0, 1.0, 5.0, -0.0, 1e-45, 3.0, 0.5]
    mfd = np.array([k6.refk_mul_float_double(float(p), float(q)) for p, q in zip(a, b)], dtype=np.float64)
    mff = np.array([k6.refk_mul_float_float(float(p), float(q)) for p, q in zip(a, b)], dtype=np.float32)
    iv = rng.integers(-2**31, 2**31, n, dtype=np.int64).astype(np.int32)
    iv[:12] = [0, 1, -1, 2**31 - 1, -2**31, -2**31 + 1, 255, 256, 65535, 65536, 2**24, 2**24 + 1]
    small = iv >> rng.integers(0, 31, n).astype(np.int32)
    iv = np.concatenate([iv, small]).astype(np.int32)
    itf = np.array([k6.refk_int_to_float_scaled(int(v), 31) for v in iv], dtype=np.float32)
    itd = np.array([k6.refk_int_to_double_scaled(int(v), 31) for v in iv], dtype=np.float64)
    # doubles in the range where dsps31Double0DB is defined (|d| >= 2^-42) plus saturating ones
    dv = rng.standard_normal(n) * np.exp2(rng.integers(-40, 3, n))
    dv[:6] = [0.0, 1.0, -1.0, 0.999999999, -0.999999999, 2.5]
    s31 = np.array([k6.refk_s31_from_double(float(v)) for v in dv], dtype=np.int32)
    satd = np.array([k6.refk_saturate_double(float(v)) for v in dv], dtype=np.float64)
    trd = np.array([k6.refk_truncate_double(float(v), 24) for v in dv], dtype=np.float64)
    lv = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    lv[:6] = [0, 2**59, 2**59 - 1, -2**59, -2**59 - 1, 12345 << 28]
    sat64 = np.array([k2.refk_saturate64_031(int(v), 28) for v in lv], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "kernel_vectors.npz"), a=a, b=b, mul_float_double=mfd,
                        mul_float_float=mff, iv=iv, int_to_float_scaled=itf, int_to_double_scaled=itd,
                        dv=dv, s31_from_double=s31, saturate_double=satd, truncate_double_24=trd,
                        lv=lv, saturate64_031=sat64)

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(dict(cases=manifest, reference_encoder=enc, dspcreate_reproduces=repro,
                       init_return_codes=neg,
                       note="generated by tests/golden/make_goldens.py from the compiled reference "
                            "(oracle/_ref); inputs are pb.lcg_input / explicit arrays in that script"),
                  f, indent=1)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {len(manifest)} cases, {total / 1024:.0f} KiB under tests/golden/")


if __name__ == "__main__":
    main()
