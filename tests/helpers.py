"""Shared helpers for the parity tests (test infrastructure)."""
import importlib

import numpy as np

synth = importlib.import_module("x-slam_amd.synth")


def cmat(real, imag=None):
    """float64 real (+ optional imag) matrix -> float32 array with trailing (re, im)."""
    real = np.asarray(real, np.float64)
    out = np.zeros(real.shape + (2,), np.float32)
    out[..., 0] = real
    if imag is not None:
        out[..., 1] = imag
    return out


def s1_transforms(k, prm, seed=(0, 3), h=1e-7):
    """Complex transforms the orchestrator hands to the kernels for frame k of scene S1,
    computed in float64 and rounded once (inputs only — both sides receive the same bits).
    The CSFD seed i*h sits on world2camera(seed) as in KinectFusionReconstruction.cpp:22."""
    c2w = synth.s1_pose(k).astype(np.complex128)
    w2c = np.linalg.inv(c2w)
    if seed is not None:
        w2c[seed] += 1j * h
    c2w = np.linalg.inv(w2c)
    w2v = np.eye(4, dtype=np.complex128)
    w2v[:3, 3] = [prm["init_x"], prm["init_y"], prm["init_z"]]
    c2v = w2v @ c2w
    v2c = np.linalg.inv(c2v)
    v2w = np.linalg.inv(w2v)
    f = lambda m: cmat(m.real, m.imag)
    return dict(Rv2c=f(v2c[:3, :3]), tv2c=f(v2c[:3, 3]), Rc2v=f(c2v[:3, :3]), tc2v=f(c2v[:3, 3]),
                Rv2w=f(v2w[:3, :3]), tv2w=f(v2w[:3, 3]), Rc2w=f(c2w[:3, :3]), tc2w=f(c2w[:3, 3]), c2w=c2w, w2c=w2c)


def tranc_dist(prm):
    vs = np.float32(prm["tsdf_voxel_size"])
    return float(max(np.float32(vs * np.float32(prm["thres_range"])), np.float32(np.float32(2.1) * vs)))


def intr_of(prm, level=0):
    d = np.float32(1 << level)
    return np.array([np.float32(prm["fx"]) / d, np.float32(prm["fy"]) / d, np.float32(prm["cx"]) / d, np.float32(prm["cy"]) / d],
                    np.float32)


def mismatch_fraction(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    same = (a == b) | (np.isnan(a) & np.isnan(b)) if a.dtype.kind == "f" else (a == b)
    return 1.0 - same.mean()
