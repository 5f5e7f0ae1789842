"""Shared helpers for the parity tests."""
import numpy as np

SEED = 20260424
HFOV_12MM = 112.61986494804043   # default preset, f=12mm on 36mm
HFOV_14MM = 104.2500326978036
HFOV_17MM = 93.27315296578

TEMPLATE_CALIB = dict(width=3840, height=3840, f=1049.9268186384606, cx=-0.053481903280599763,
                      cy=-0.040449115818567277, k1=0.10190869149858893, k2=0.00079808296648272998,
                      k3=-0.00031893309097734927)
FULL_CALIB = dict(width=640, height=480, f=170.25, cx=3.5, cy=-2.25, k1=0.08, k2=-0.01, k3=0.002, k4=-0.0003,
                  p1=0.0007, p2=-0.0004, b1=1.75, b2=-0.6)


def rand_image(h, w, c=3, seed=SEED):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w, c), dtype=np.uint8)


def norm_yaw(a):
    a = ((a + 180.0) % 360.0) - 180.0
    return 180.0 if abs(a + 180.0) < 1e-6 else a


def ring_views(count, size, hfov, pitch=0.0):
    """(yaw, pitch, hfov, vfov, w, h) tuples of an N-view ring (PC:794)."""
    return [(norm_yaw(i * 360.0 / count), pitch, hfov, hfov, size, size) for i in range(count)]


PRESET_FULL360 = [(0, 0), (45, 30), (45, -30), (90, 0), (135, 30), (135, -30), (180, 0), (-135, 30), (-135, -30),
                  (-90, 0), (-45, 30), (-45, -30)]
PRESET_FISHEYELIKE = [(0, 0), (0, 30), (0, -30), (36, 0), (144, 0), (180, 0), (180, 30), (180, -30), (-144, 0), (-36, 0)]
