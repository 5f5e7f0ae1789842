"""Image file decode/encode for the drop-in CLIs.

File codecs are outside the measured hot path.  Pillow is used when importable; a
self-contained 8-bit PNG reader/writer (zlib) keeps the tools usable without it.  Arrays are H x W x C uint8
in RGB(A) order.
"""
import pathlib
import struct
import zlib

import numpy as np

try:  # Pillow is present in the ROCm image but is not a hard dependency
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
except Exception:  # pragma: no cover
    Image = None


class ImageIOError(RuntimeError):
    pass


# ---- minimal PNG ---------------------------------------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"
_COLOR_CH = {0: 1, 2: 3, 4: 2, 6: 4}


def _png_read(data: bytes) -> np.ndarray:
    if data[:8] != _PNG_SIG:
        raise ImageIOError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        pos += 12 + ln
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if hdr is None:
        raise ImageIOError("PNG without IHDR")
    w, h, depth, ctype, _comp, _flt, interlace = hdr
    if depth != 8 or ctype not in _COLOR_CH or ctype == 4 or interlace:
        raise ImageIOError("built-in PNG reader handles 8-bit gray/RGB/RGBA non-interlaced only (install Pillow)")
    ch = _COLOR_CH[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    stride = w * ch
    rows = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft = int(rows[y, 0])
        line = rows[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.copy()
            for c in range(ch):  # per-channel running sum mod 256
                cur[c::ch] = np.cumsum(line[c::ch]) & 255
        else:  # average / paeth: scalar loop (rare for synthetic data, correct if slow)
            cur = np.zeros(stride, np.int32)
            for i in range(stride):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c0 = prev[i - ch] if i >= ch else 0
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c0
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c0)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c0)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, ch)


def _png_write(path: pathlib.Path, arr: np.ndarray, level: int = 3) -> None:
    h, w, ch = arr.shape
    ctype = {1: 0, 3: 2, 4: 6}[ch]

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)

    raw = np.zeros((h, w * ch + 1), np.uint8)
    raw[:, 1:] = arr.reshape(h, w * ch)
    data = _PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) \
        + chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + chunk(b"IEND", b"")
    path.write_bytes(data)


# ---- public --------------------------------------------------------------------------------------
def read_image(path) -> np.ndarray:
    """-> H x W x C uint8 (C = 1, 3 or 4; RGB order)."""
    path = pathlib.Path(path)
    if Image is not None:
        try:
            with Image.open(path) as im:
                if im.mode in ("I;16", "I;16B", "I;16L", "I"):
                    a = np.asarray(im).astype(np.float32)
                    a = (a / 257.0 + 0.5).clip(0, 255).astype(np.uint8)   # 16-bit sources: u8 path only
                    return a[:, :, None]
                if im.mode not in ("L", "RGB", "RGBA"):
                    im = im.convert("RGBA" if "A" in im.mode else "RGB")
                a = np.asarray(im)
        except Exception as exc:
            raise ImageIOError(f"cannot read {path}: {exc}") from exc
        return np.ascontiguousarray(a if a.ndim == 3 else a[:, :, None])
    if path.suffix.lower() == ".png":
        return _png_read(path.read_bytes())
    raise ImageIOError(f"cannot read {path}: Pillow is not installed and the built-in codec is PNG only")


def write_image(path, arr: np.ndarray, jpeg_q: int = None) -> None:
    """Encode by extension.  jpeg_q is ffmpeg's -q:v (1 = best, 2 ~ 95 %), mapped onto Pillow qualities
    with 4:4:4 sampling and optimised Huffman tables like the reference's mjpeg flags (PC:331-338)."""
    path = pathlib.Path(path)
    ext = path.suffix.lower()
    a = np.ascontiguousarray(arr)
    if a.ndim == 2:
        a = a[:, :, None]
    path.parent.mkdir(parents=True, exist_ok=True)
    if Image is not None:
        mode = {1: "L", 3: "RGB", 4: "RGBA"}[a.shape[2]]
        im = Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a, mode)
        if ext in (".jpg", ".jpeg"):
            if mode == "RGBA":
                im = im.convert("RGB")
            quality = 95 if (jpeg_q is not None and jpeg_q >= 2) else 100
            im.save(path, "JPEG", quality=quality, subsampling=0, optimize=True)
        elif ext == ".png":
            im.save(path, "PNG", compress_level=3)
        elif ext in (".tif", ".tiff"):
            im.save(path, "TIFF")
        else:
            raise ImageIOError(f"unsupported output extension {ext}")
        return
    if ext == ".png":
        _png_write(path, a)
        return
    raise ImageIOError(f"cannot write {path}: Pillow is not installed and the built-in codec is PNG only")
