"""Shared helpers for the parity tests (test infrastructure)."""
import importlib

import numpy as np

synth = importlib.import_module("x-slam_amd.synth")


cmat, s1_transforms, tranc_dist, intr_of = synth.cmat, synth.s1_transforms, synth.tranc_dist, synth.intr_of


def mismatch_fraction(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    same = (a == b) | (np.isnan(a) & np.isnan(b)) if a.dtype.kind == "f" else (a == b)
    return 1.0 - same.mean()
