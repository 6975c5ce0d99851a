"""Synthetic clouds of BASELINE.json's configs 4 and 5.

Counter-based, so every rank and the CPU baseline generate identical points without shared RNG
state:  u = splitmix64(seed ^ (3*i + k));  coord = ((u >> 40) * 2^-24) * 2 - 1   in [-1, 1).
The cloud to move is P = R_g M + t_g evaluated in fp32 exactly as apply_alignment does
(reference libcpu/libcpu/icp.cc:66-74), R_g = rotation by `angle_deg` about (1,2,3)/sqrt(14),
t_g = (0.05, -0.03, 0.02).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def cloud(n: int, seed: int = 12345, begin: int = 0, end: int | None = None,
          chunk: int = 1 << 22) -> np.ndarray:
    """Rows [begin, end) of the n-point uniform cloud in [-1,1)^3, float32 (n,3)."""
    end = n if end is None else end
    out = np.empty((end - begin, 3), np.float32)
    for lo in range(begin, end, chunk):
        hi = min(end, lo + chunk)
        idx = np.arange(lo, hi, dtype=np.uint64) * np.uint64(3)
        for k in range(3):
            u = _splitmix64(np.uint64(seed) ^ (idx + np.uint64(k)))
            v = (u >> np.uint64(40)).astype(np.float64) * (2.0 ** -24)
            out[lo - begin:hi - begin, k] = (v * 2.0 - 1.0).astype(np.float32)
    return out


def ground_truth(angle_deg: float = 1.0) -> np.ndarray:
    """4x4 float32 [R_g t_g; 0 0 0 1]."""
    a = np.deg2rad(angle_deg)
    ax = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = [0.05, -0.03, 0.02]
    return T.astype(np.float32)


def apply_fp32(p: np.ndarray, T: np.ndarray) -> np.ndarray:
    """apply_alignment's arithmetic (icp.cc:66-74): ((x*T0 + y*T1) + z*T2) + T3, all fp32."""
    p = np.ascontiguousarray(p, np.float32)
    T = np.asarray(T, np.float32)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = ((x * T[r, 0] + y * T[r, 1]) + z * T[r, 2]) + T[r, 3]
    return out


def expected_transform(angle_deg: float = 1.0) -> np.ndarray:
    """The transform ICP should recover: the inverse of the ground truth (float64)."""
    return np.linalg.inv(ground_truth(angle_deg).astype(np.float64))
