"""Input colour stage of the dual-fisheye tool on the GPU: .cube loader and the host half of the colour plan.

Replaces `apply_input_color_pipeline` (reference cli_tools/gs360_DualFisheyeDistortionCalibration.py:684-725, called
per image from load_prepared_input_image DF:728-743) for 8-bit images (below) and 16-bit images (`output_pieces16`: the
input side is evaluated per pixel in the kernel, only the implementation-defined sRGB re-encode is tabulated, in the three
pieces on which it is monotone at 16-bit resolution).  An 8-bit image has 256 input levels per
channel and 256 output levels, so both scalar ends of the pipeline are tabulated here with the reference's own
float32 NumPy expressions and handed to the kernel (include/gs360.h, "input colour stage"):

  level_positions()    level v -> float01 -> `clip((x - domain_min) / span, 0, 1) * (size - 1)`   (DF:603-613, 647-649)
  output_thresholds()  the encode step `LUT output x -> uint8` (optional Rec.709 -> sRGB re-encode DF:565-600, then
                       `rint(clip(x) * 255)` DF:616-618) is a monotone step function of x; its 255 step positions are
                       found by bisection over float32 bit patterns using this host's NumPy -- so the GPU result is the
                       byte NumPy would produce here, including NumPy's own (implementation-defined) float32 power.

The 8-texel trilinear interpolation between the two ends runs in the kernel.  No CPU path: applying a stage needs a
gs360 Context.
"""
import dataclasses
import pathlib
import threading
from typing import Dict, Optional

import numpy as np

F32 = np.float32
_ONE_BITS = 0x3F800000          # float32 1.0


@dataclasses.dataclass
class CubeLUT:                   # same fields as the reference's container (DF:88-96)
    size: int
    table: np.ndarray            # float32 [b][g][r][3]
    domain_min: np.ndarray       # float32 [3]
    domain_max: np.ndarray


def normalize_lut_output_color_space(value) -> str:
    """`native` is the legacy spelling of `passthrough` (DF:480-491)."""
    text = str(value or "passthrough").strip().lower()
    if text == "native":
        text = "passthrough"
    if text not in ("passthrough", "srgb"):
        raise ValueError("Unsupported --lut-output-color-space: {}".format(value))
    return text


def load_cube_lut(lut_path) -> CubeLUT:
    """Parse a .cube 3D LUT; accepts/rejects what the reference loader does, with its messages (DF:494-562)."""
    lut_path = pathlib.Path(lut_path)
    if not lut_path.is_file():
        raise FileNotFoundError("LUT file not found: {}".format(lut_path))
    size = None
    domain = {"DOMAIN_MIN": [0.0, 0.0, 0.0], "DOMAIN_MAX": [1.0, 1.0, 1.0]}
    triples = []
    with lut_path.open("r", encoding="utf-8", errors="ignore") as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line[0] == "#":
                continue
            head = line.upper()
            if head.startswith("TITLE"):
                continue
            fields = line.split()
            if head.startswith("LUT_3D_SIZE"):
                if len(fields) < 2:
                    raise ValueError("Invalid LUT_3D_SIZE line: {}".format(line))
                size = int(fields[1])
            elif head.startswith("DOMAIN_MIN") or head.startswith("DOMAIN_MAX"):
                key = head[:10]
                if len(fields) != 4:
                    raise ValueError("Invalid {} line: {}".format(key, line))
                domain[key] = [float(t) for t in fields[1:]]
            elif len(fields) == 3:
                triples.append((float(fields[0]), float(fields[1]), float(fields[2])))
    if size is None:
        raise ValueError("LUT_3D_SIZE is missing in {}".format(lut_path))
    if size <= 1:
        raise ValueError("LUT_3D_SIZE must be > 1 in {}".format(lut_path))
    if len(triples) != size ** 3:
        raise ValueError("LUT row count mismatch in {}: got {}, expected {}".format(lut_path, len(triples), size ** 3))
    dmin, dmax = np.array(domain["DOMAIN_MIN"], F32), np.array(domain["DOMAIN_MAX"], F32)
    if np.any(dmax - dmin <= 0.0):
        raise ValueError("Invalid LUT domain range in {}".format(lut_path))
    return CubeLUT(size, np.asarray(triples, F32).reshape(size, size, size, 3), dmin, dmax)


# ---- host half of the plan ---------------------------------------------------------------------------------------
def level_positions(lut: CubeLUT) -> np.ndarray:
    """float32 [3][256]: LUT-grid position of each 8-bit level per channel (R, G, B)."""
    x = np.arange(256, dtype=np.uint8).astype(F32) / 255.0                     # DF:606-607
    span = (lut.domain_max - lut.domain_min).astype(F32)
    coord = np.clip((x[:, None] - lut.domain_min[None, :]) / span[None, :], 0.0, 1.0)   # DF:647
    pos = coord * float(lut.size - 1)                                          # DF:648
    return np.ascontiguousarray(pos.T.astype(F32))


def encode_levels(x: np.ndarray, space: str) -> np.ndarray:
    """float32 LUT output -> uint8 level, the tail of the pipeline (DF:707-716), in NumPy float32."""
    v = np.clip(np.asarray(x, F32), 0.0, 1.0)
    if space == "srgb":
        with np.errstate(invalid="ignore"):
            lin = np.where(v < 0.081, v / 4.5, np.power((v + 0.099) / 1.099, 1.0 / 0.45)).astype(F32)     # DF:568-574
            lin = np.clip(lin, 0.0, 1.0)
            enc = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(lin, 1.0 / 2.4) - 0.055).astype(F32)  # DF:580-587
        v = np.clip(np.clip(enc, 0.0, 1.0), 0.0, 1.0)
    elif space != "passthrough":
        raise ValueError("Unexpected LUT output color space")
    return np.rint(v * 255.0).astype(np.uint8)                                 # DF:616-618


def _bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return np.asarray(bits, np.int64).astype(np.uint32).view(F32)


def _levels_from_thresholds(thr: np.ndarray, x: np.ndarray) -> np.ndarray:
    return np.searchsorted(thr[1:], x, side="right").astype(np.uint8)          # count of thresholds <= x


def output_thresholds(space: str, verify: bool = True) -> np.ndarray:
    """float32 [256]; entry k (1..255) is the smallest float32 x in [0,1] with encode_levels(x) >= k, +inf if none."""
    space = normalize_lut_output_color_space(space)
    k = np.arange(1, 256)
    lo = np.full(255, -1, np.int64)                  # encode(lo) < k   (bit pattern -1 = "below 0.0")
    hi = np.full(255, _ONE_BITS + 1, np.int64)       # encode(hi) >= k  (one past 1.0 = "never")
    while np.any(hi - lo > 1):
        mid = (lo + hi) >> 1
        ge = encode_levels(_bits_to_f32(np.clip(mid, 0, _ONE_BITS)), space) >= k
        hi = np.where(ge, mid, hi)
        lo = np.where(ge, lo, mid)
    thr = np.empty(256, F32)
    thr[0] = -np.inf
    thr[1:] = np.where(hi > _ONE_BITS, np.inf, _bits_to_f32(np.clip(hi, 0, _ONE_BITS)))
    if verify:
        _verify_monotone(thr, space)
    return thr


def _verify_monotone(thr: np.ndarray, space: str) -> None:
    """The table is exact iff the encode step is monotone; check it where it could fail on this host's NumPy."""
    probes = [_bits_to_f32(np.linspace(0, _ONE_BITS, 1 << 20).astype(np.int64))]
    for centre in (0.081, 0.0031308 * 4.5, 0.018, 1.0):      # the two piecewise joints (encoded and linear side), the top
        c = int(np.array(centre, F32).view(np.uint32))
        probes.append(_bits_to_f32(np.clip(np.arange(c - 20000, c + 20000), 0, _ONE_BITS)))
    finite = thr[1:][np.isfinite(thr[1:])]
    fb = finite.view(np.uint32).astype(np.int64)
    probes.append(_bits_to_f32(np.clip(np.concatenate([fb - 1, fb, fb + 1]), 0, _ONE_BITS)))
    x = np.concatenate(probes)
    if not np.array_equal(_levels_from_thresholds(thr, x), encode_levels(x, space)):
        raise RuntimeError("colour encode step ({}) is not monotone under this NumPy; refusing to tabulate it".format(space))


# ---- 16-bit images ---------------------------------------------------------------------------------------------------
def encode_levels16(x: np.ndarray, space: str) -> np.ndarray:
    """float32 LUT output -> uint16 level (DF:707-716 with float01_to_image's uint16 branch, DF:623-624)."""
    v = np.clip(np.asarray(x, F32), 0.0, 1.0)
    if space == "srgb":
        with np.errstate(invalid="ignore"):
            lin = np.where(v < 0.081, v / 4.5, np.power((v + 0.099) / 1.099, 1.0 / 0.45)).astype(F32)
            lin = np.clip(lin, 0.0, 1.0)
            enc = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(lin, 1.0 / 2.4) - 0.055).astype(F32)
        v = np.clip(np.clip(enc, 0.0, 1.0), 0.0, 1.0)
    elif space != "passthrough":
        raise ValueError("Unexpected LUT output color space")
    return np.rint(v * 65535.0).astype(np.uint16)


def _first_true(pred) -> int:
    """smallest float32 bit pattern in [0, bits(1.0)] for which the monotone predicate holds (bits(1.0)+1 if never)"""
    lo, hi = -1, _ONE_BITS + 1
    while hi - lo > 1:
        mid = (lo + hi) >> 1
        if bool(pred(_bits_to_f32(np.array([mid]))[0])):
            hi = mid
        else:
            lo = mid
    return hi


def output_pieces16(space: str, verify: bool = True):
    """The 16-bit output quantiser as data for gs360_color_plan16_create: (n_pieces, start[4], base[4], off[5], thresholds).

    `passthrough` needs no table (n_pieces = 0: the kernel evaluates rint(clip(x) * 65535) itself).  The sRGB re-encode is
    tabulated with THIS host's NumPy (its float32 power is implementation-defined) in the three pieces on which it is
    monotone: Rec.709 toe below the sRGB toe, Rec.709 toe above it, and the power segment (v >= 0.081)."""
    space = normalize_lut_output_color_space(space)
    start = np.zeros(4, F32)
    base = np.zeros(4, np.int32)
    off = np.zeros(5, np.int32)
    if space == "passthrough":
        return 0, start, base, off, np.zeros(0, F32)
    with np.errstate(invalid="ignore"):
        b1 = _first_true(lambda v: not ((F32(v) / F32(4.5)) <= F32(0.0031308)))      # first v past the sRGB toe (DF:583)
        b2 = _first_true(lambda v: not (F32(v) < F32(0.081)))                        # first v on the power segment (DF:571)
    if not (0 < b1 < b2 <= _ONE_BITS):
        raise RuntimeError("unexpected piece boundaries of the Rec.709 -> sRGB re-encode")
    bounds = [0, b1, b2, _ONE_BITS + 1]
    thr_all = []
    for p in range(3):
        a, b = bounds[p], bounds[p + 1] - 1                      # inclusive bit range of the piece
        kmin = int(encode_levels16(_bits_to_f32(np.array([a])), space)[0])
        kmax = int(encode_levels16(_bits_to_f32(np.array([b])), space)[0])
        if kmax < kmin:
            raise RuntimeError("colour encode step decreases over piece {} under this NumPy".format(p))
        k = np.arange(kmin + 1, kmax + 1)
        lo = np.full(k.size, a - 1, np.int64)                    # encode(lo) < k
        hi = np.full(k.size, b, np.int64)                        # encode(hi) >= k  (true at the piece's top for every k <= kmax)
        while np.any(hi - lo > 1):
            mid = (lo + hi) >> 1
            ge = encode_levels16(_bits_to_f32(mid), space).astype(np.int64) >= k
            hi = np.where(ge, mid, hi)
            lo = np.where(ge, lo, mid)
        thr = _bits_to_f32(hi)
        if verify:
            probes = [np.linspace(a, b, 1 << 18).astype(np.int64), np.clip(np.concatenate([hi - 1, hi, hi + 1]), a, b)]
            x = _bits_to_f32(np.concatenate(probes))
            got = kmin + np.searchsorted(thr, x, side="right")
            if not np.array_equal(got, encode_levels16(x, space).astype(np.int64)):
                raise RuntimeError("colour encode step ({}) is not monotone on piece {} under this NumPy; refusing to "
                                   "tabulate it".format(space, p))
        start[p] = _bits_to_f32(np.array([a]))[0]
        base[p] = kmin
        off[p + 1] = off[p] + thr.size
        thr_all.append(thr)
    return 3, start, base, off, np.ascontiguousarray(np.concatenate(thr_all), F32)


class ColorStage:
    """A loaded LUT + output colour space, applied to device-resident 8- or 16-bit images through the C ABI."""

    def __init__(self, lut: CubeLUT, output_space: str = "srgb"):
        self.lut = lut
        self.space = normalize_lut_output_color_space(output_space)
        self.level_pos = level_positions(lut)
        self.thresholds = output_thresholds(self.space)
        self._plans: Dict[object, object] = {}      # Context -> plan handle (the key keeps the context object alive)
        self._plans16: Dict[object, object] = {}    # the 16-bit plans (built on first use: their table takes a second)
        self._pieces16 = None
        self._lock = threading.Lock()

    def _plan(self, ctx):
        with self._lock:
            plan = self._plans.get(ctx)
            if plan is None:
                plan = self._plans[ctx] = ctx.color_plan(self.lut.table, self.level_pos, self.thresholds)
            return plan

    def _plan16(self, ctx):
        with self._lock:
            plan = self._plans16.get(ctx)
            if plan is None:
                if self._pieces16 is None:
                    self._pieces16 = output_pieces16(self.space)
                plan = self._plans16[ctx] = ctx.color_plan16(self.lut.table, self.lut.domain_min, self.lut.domain_max, *self._pieces16)
            return plan

    @staticmethod
    def check_image(shape, dtype) -> None:
        if len(shape) < 3 or shape[2] < 3:
            raise ValueError("LUT-based input conversion requires at least 3-channel RGB image input")   # DF:693-697
        if np.dtype(dtype) not in (np.dtype(np.uint8), np.dtype(np.uint16)):
            raise TypeError("the gs360 colour stage handles 8- and 16-bit integer images (got {})".format(np.dtype(dtype)))
        if shape[2] > 4:
            raise ValueError("images with more than 4 channels are not supported")

    def apply_dev(self, ctx, buf, shape, red_index: int = 0, slot: int = 0, dtype=np.uint8) -> None:
        """In place on a device buffer holding an H x W x C uint8 / uint16 image (C = 3 or 4; alpha is kept)."""
        self.check_image(shape, dtype)
        if np.dtype(dtype) == np.uint16:
            ctx.color_apply16_dev(self._plan16(ctx), buf, int(shape[0]), int(shape[1]), int(shape[2]), red_index=red_index, slot=slot)
            return
        ctx.color_apply_dev(self._plan(ctx), buf, int(shape[0]), int(shape[1]), int(shape[2]), red_index=red_index, slot=slot)

    def apply(self, ctx, image: np.ndarray, red_index: int = 0, slot: int = 0) -> np.ndarray:
        """Host convenience: upload, convert, download.  `red_index=2` for BGR(A) arrays (cv2.imread order)."""
        self.check_image(image.shape, image.dtype)
        img = np.ascontiguousarray(image)
        with ctx.slot_locks[slot]:
            d = ctx.to_device(img, slot=slot)
            try:
                self.apply_dev(ctx, d, img.shape, red_index=red_index, slot=slot, dtype=img.dtype)
                return ctx.download(d, img.shape, dtype=img.dtype, slot=slot)
            finally:
                ctx.free(d)

    def close(self) -> None:
        with self._lock:
            for ctx, plan in self._plans.items():
                if ctx.handle:
                    ctx.color_plan_free(plan)
            self._plans.clear()
            for ctx, plan in self._plans16.items():
                if ctx.handle:
                    ctx.color_plan16_free(plan)
            self._plans16.clear()


def make_stage(lut_path, output_space: str = "srgb") -> Optional[ColorStage]:
    return ColorStage(load_cube_lut(lut_path), output_space) if lut_path else None
