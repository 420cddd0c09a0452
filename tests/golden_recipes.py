"""Program / input recipes shared by tests/golden/make_goldens.py (which runs the compiled reference
on them) and the parity tests (which re-create the same programs and inputs on the GPU box, where
the reference does not exist).  Pure numpy; nothing here touches the oracle or the reference."""
from __future__ import annotations

import os

import numpy as np

from avdsp_amd import progbuilder as pb

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def is_float_samples(fmt: int) -> bool:
    return fmt in (5, 6)


def make_program(recipe: dict) -> np.ndarray:
    kind = recipe["kind"]
    if kind == "file":
        return np.fromfile(os.path.join(GOLDEN_DIR, recipe["name"]), dtype=np.uint32)
    if kind == "synth":
        return pb.synth_program(recipe["fmt"], recipe["channels"], recipe["sections"],
                                recipe.get("taps", 0), recipe.get("fmin", pb.F48000),
                                recipe.get("fmax", pb.F48000), recipe.get("gain", 1.0))
    if kind == "fuzz":                 # random well-formed program, see tests/fuzz_programs.py
        from tests.fuzz_programs import random_program
        return random_program(recipe["seed"], recipe["fmt"])
    raise ValueError(kind)


def make_input(recipe: dict, fmt: int) -> np.ndarray:
    fl = is_float_samples(fmt)
    frames, ch = recipe["frames"], recipe["channels"]
    kind = recipe["kind"]
    if kind == "lcg":
        return pb.lcg_input(frames, ch, fl, seed=recipe.get("seed", 12345))
    if kind == "fullscale":            # +/- (almost) full scale square-ish pattern: drives every saturator
        t = np.arange(frames * ch).reshape(frames, ch)
        sq = np.sign(np.sin(t * 0.37))
        return (sq * 0.999).astype(np.float32) if fl else (sq * 2147483000).astype(np.int32)
    if kind == "denormals":            # subnormal / barely-normal samples, both signs: the FTZ/DAZ corner of the reference's build
        x = np.zeros((frames, ch), dtype=np.float32 if fl else np.int32)
        vals_f = [1e-40, 3e-39, 1.5e-38, -2e-41, 1e-37, -1.2e-38, 5e-45, -1e-30]
        vals_i = [1, -3, 100, -1, 7, -100, 2, -2]
        for k in range(min(frames, 24)):
            x[k, k % ch] = vals_f[k % 8] if fl else vals_i[k % 8]
        return x
    if kind == "impulse":
        x = np.zeros((frames, ch), dtype=np.float32 if fl else np.int32)
        x[0, :] = recipe["value_f"] if fl else recipe["value_i"]
        return x
    raise ValueError(kind)



def check_against_golden(case: dict, out: np.ndarray, state: np.ndarray, sha) -> None:
    """Outputs and final state area against what the compiled reference produced, bit for bit."""
    g = np.load(os.path.join(GOLDEN_DIR, case["name"] + ".npz"))

    def same(a, b, what):
        a, b = a.view(np.uint32), b.view(np.uint32)
        assert (a == b).all(), f"{case['name']}: {what} differs from the reference's " \
            f"(columns {sorted(set(np.nonzero(a != b)[1].tolist()))[:8]})"

    same(out[:16], g["head"], "head")
    same(out[-16:], g["tail"], "tail")
    if case["full"]:
        same(out, g["out"], "output")
        assert (state == g["state"]).all(), f"{case['name']}: state area differs from the reference's"
    assert sha(out) == case["out_sha"]
    assert sha(state) == case["state_sha"]
