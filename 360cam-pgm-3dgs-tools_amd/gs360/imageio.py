"""Image file decode/encode for the drop-in CLIs.

File codecs are outside the measured hot path.  Arrays are H x W x C in RGB(A) order, uint8 or --
for 16-bit sources, which the reference keeps at native depth (DF:735 cv2.imread(IMREAD_UNCHANGED); PC:327-347 writes no
-pix_fmt for PNG/TIFF stills and rgb48le for > 8-bit videos) -- uint16.

8-bit files go through Pillow when it is importable (a self-contained 8-bit PNG codec keeps the tools usable without it).
16-bit files never go through Pillow (it delivers 16-bit RGB as 8-bit): PNG is decoded here (zlib + the library's
gs360_png_unfilter helper) and written with filter type 0; TIFF is read / written as baseline TIFF (uncompressed,
LZW or Deflate, chunky, strips, predictor 1 or 2) -- anything else 16-bit raises ImageIOError rather than losing depth.
"""
import os
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


# ---- PNG -----------------------------------------------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"
_COLOR_CH = {0: 1, 2: 3, 4: 2, 6: 4}


def _png_header(data: bytes):
    if data[:8] != _PNG_SIG or data[12:16] != b"IHDR":
        raise ImageIOError("not a PNG file")
    return struct.unpack(">IIBBBBB", data[16:29])      # w, h, depth, colour type, compression, filter, interlace


def _unfilter_py(rows: np.ndarray, h: int, stride: int, bpp: int) -> None:
    """pure-NumPy fallback of gs360_png_unfilter (row loop; average / Paeth rows fall to a scalar loop)"""
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
            for c in range(bpp):
                cur[c::bpp] = np.cumsum(line[c::bpp]) & 255
        elif ft in (3, 4):
            cur = np.zeros(stride, np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c0 = prev[i - bpp] if i >= bpp else 0
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c0
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c0)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c0)
                cur[i] = (line[i] + pred) & 255
        else:
            raise ImageIOError(f"PNG row {y} has unknown filter type {ft}")
        rows[y, 1:] = cur
        prev = cur


def _png_read(data: bytes) -> np.ndarray:
    w, h, depth, ctype, _comp, _flt, interlace = _png_header(data)
    pos, idat = 8, []
    while pos + 8 <= len(data):
        ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
        if pos + 12 + ln > len(data):
            raise ImageIOError(f"truncated PNG: chunk {typ!r} at byte {pos} runs past the end of the file")
        if typ == b"IDAT":
            body = data[pos + 8:pos + 8 + ln]
            if zlib.crc32(typ + body) & 0xFFFFFFFF != struct.unpack(">I", data[pos + 8 + ln:pos + 12 + ln])[0]:
                raise ImageIOError(f"corrupt PNG: CRC mismatch in the IDAT chunk at byte {pos}")
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + ln
    if depth not in (8, 16) or ctype not in (0, 2, 4, 6) or interlace:
        raise ImageIOError("built-in PNG reader handles non-interlaced 8/16-bit gray, gray+alpha, RGB and RGBA; convert palette / "
                           "interlaced / 1-2-4-bit files first (e.g. `magick in.png -interlace none -depth 16 out.png`)")
    ch = _COLOR_CH[ctype]
    bpp = ch * depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    if raw.size != h * (stride + 1):
        raise ImageIOError("PNG data size does not match its header")
    rows = raw.reshape(h, stride + 1).copy()
    done = False
    try:
        from . import capi
        L = capi.load_library()
        done = L.gs360_png_unfilter(rows.ctypes.data, h, stride, bpp) == 0
    except Exception:  # noqa: BLE001  (library not built: slow path)
        done = False
    if not done:
        rows = raw.reshape(h, stride + 1).copy()
        _unfilter_py(rows, h, stride, bpp)
    px = np.ascontiguousarray(rows[:, 1:])
    img = px.view(">u2").astype(np.uint16).reshape(h, w, ch) if depth == 16 else px.reshape(h, w, ch)
    if ctype == 4:                           # gray + alpha -> 4 channels, as cv2.imread(IMREAD_UNCHANGED) delivers it (DF:735)
        img = np.ascontiguousarray(np.concatenate([np.repeat(img[:, :, :1], 3, axis=2), img[:, :, 1:]], axis=2))
    return img


def _png_write(path: pathlib.Path, arr: np.ndarray, level: int = 3) -> None:
    h, w, ch = arr.shape
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    depth = 16 if arr.dtype == np.uint16 else 8

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)

    body = arr.astype(">u2").view(np.uint8).reshape(h, w * ch * 2) if depth == 16 else arr.reshape(h, w * ch)
    raw = np.zeros((h, body.shape[1] + 1), np.uint8)
    raw[:, 1:] = body
    data = _PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) \
        + chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + chunk(b"IEND", b"")
    path.write_bytes(data)


# ---- baseline TIFF (16-bit path) --------------------------------------------------------------------
_TIFF_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 16: "Q"}


def _tiff_tags(data: bytes):
    if data[:2] == b"II":
        e = "<"
    elif data[:2] == b"MM":
        e = ">"
    else:
        raise ImageIOError("not a TIFF file")
    if struct.unpack(e + "H", data[2:4])[0] != 42:
        raise ImageIOError("BigTIFF / unknown TIFF flavour is not supported")
    off = struct.unpack(e + "I", data[4:8])[0]
    n = struct.unpack(e + "H", data[off:off + 2])[0]
    tags = {}
    for k in range(n):
        tag, typ, cnt, val = struct.unpack(e + "HHI4s", data[off + 2 + 12 * k:off + 14 + 12 * k])
        if typ not in _TIFF_TYPES:
            continue
        fmt = _TIFF_TYPES[typ]
        size = struct.calcsize(fmt) * cnt
        blob = val[:size] if size <= 4 else data[struct.unpack(e + "I", val)[0]:struct.unpack(e + "I", val)[0] + size]
        tags[tag] = struct.unpack(e + fmt * cnt, blob) if fmt != "c" else blob
    return e, tags


def _lzw_decode(blob: bytes, nbytes: int) -> bytes:
    """TIFF LZW strip -> bytes through the library's host helper (no pure-Python fallback: it would take minutes per image)"""
    import ctypes as C
    try:
        from . import capi
        L = capi.load_library()
    except Exception as exc:  # noqa: BLE001
        raise ImageIOError(f"LZW-compressed 16-bit TIFF needs libgs360hip.so ({exc})") from exc
    src = np.frombuffer(blob, np.uint8)
    out = np.empty(nbytes, np.uint8)
    n = C.c_size_t(0)
    if L.gs360_tiff_lzw_decode(src.ctypes.data, src.size, out.ctypes.data, out.size, C.byref(n)) != 0 or n.value != nbytes:
        raise ImageIOError("corrupt or truncated LZW strip")
    return out.tobytes()


def _tiff_read16(data: bytes) -> np.ndarray:
    e, t = _tiff_tags(data)
    w, h = t[256][0], t[257][0]
    bits = t.get(258, (1,))
    spp = t.get(277, (1,))[0]
    comp = t.get(259, (1,))[0]
    planar = t.get(284, (1,))[0]
    pred = t.get(317, (1,))[0]
    fmt = t.get(339, (1,))[0]
    if any(b != 16 for b in bits) or spp not in (1, 3, 4) or planar != 1 or fmt != 1:
        raise ImageIOError("16-bit TIFF reader handles chunky unsigned 16-bit gray / RGB / RGBA only")
    if comp not in (1, 5, 8, 32946):
        raise ImageIOError(f"16-bit TIFF with compression {comp} is not supported (use none, LZW or Deflate); refusing to reduce it to 8 bits")
    if 324 in t:
        raise ImageIOError("tiled 16-bit TIFF is not supported (rewrite it in strips, e.g. `tiffcp -s in.tif out.tif`)")
    if 273 not in t or 279 not in t or len(t[273]) != len(t[279]):
        raise ImageIOError("TIFF without consistent StripOffsets / StripByteCounts")
    offs, cnts = t[273], t[279]
    rps = t.get(278, (h,))[0]
    if rps < 1 or len(offs) * rps < h:
        raise ImageIOError(f"TIFF strips do not cover the image ({len(offs)} strips of {rps} rows for {h} rows)")
    out = np.empty((h, w * spp), np.uint16)
    y = 0
    for k, (o, c) in enumerate(zip(offs, cnts)):
        if y >= h:
            break
        if o + c > len(data):
            raise ImageIOError(f"truncated TIFF: strip {k} ({c} bytes at offset {o}) runs past the end of the file ({len(data)} bytes)")
        blob = data[o:o + c]
        rows = min(rps, h - y)
        if comp == 5:
            blob = _lzw_decode(blob, rows * w * spp * 2)
        elif comp != 1:
            try:
                blob = zlib.decompress(blob)
            except zlib.error as exc:
                raise ImageIOError(f"corrupt TIFF: strip {k} does not inflate ({exc})") from exc
        if len(blob) < rows * w * spp * 2:
            raise ImageIOError(f"corrupt TIFF: strip {k} holds {len(blob)} bytes, {rows * w * spp * 2} expected")
        a = np.frombuffer(blob, dtype=e + "u2", count=rows * w * spp).astype(np.uint16).reshape(rows, w * spp)
        if pred == 2:                        # horizontal differencing per sample
            a = np.cumsum(a.reshape(rows, w, spp).astype(np.uint32), axis=1).astype(np.uint16).reshape(rows, w * spp)
        elif pred != 1:
            raise ImageIOError(f"TIFF predictor {pred} is not supported")
        out[y:y + rows] = a
        y += rows
    return out.reshape(h, w, spp)


def _tiff_write16(path: pathlib.Path, arr: np.ndarray) -> None:
    h, w, ch = arr.shape
    body = np.ascontiguousarray(arr, dtype="<u2").tobytes()
    entries = []

    def ent(tag, typ, vals):
        entries.append((tag, typ, vals))
    ent(256, 4, [w]); ent(257, 4, [h]); ent(258, 3, [16] * ch); ent(259, 3, [1])
    ent(262, 3, [2 if ch >= 3 else 1]); ent(273, 4, [0]); ent(277, 3, [ch]); ent(278, 4, [h]); ent(279, 4, [len(body)])
    ent(284, 3, [1])
    if ch == 4:
        ent(338, 3, [2])                     # unassociated alpha
    ent(339, 3, [1] * ch)
    n = len(entries)
    ifd_off = 8
    extra_off = ifd_off + 2 + 12 * n + 4
    extra = b""
    recs = []
    for tag, typ, vals in entries:
        fmt = {3: "H", 4: "I"}[typ]
        blob = struct.pack("<" + fmt * len(vals), *vals)
        if len(blob) <= 4:
            recs.append((tag, typ, len(vals), blob.ljust(4, b"\0")))
        else:
            recs.append((tag, typ, len(vals), struct.pack("<I", extra_off + len(extra))))
            extra += blob + (b"\0" if len(blob) & 1 else b"")
    data_off = extra_off + len(extra)
    out = b"II" + struct.pack("<HI", 42, ifd_off) + struct.pack("<H", n)
    for tag, typ, cnt, val in recs:
        if tag == 273:
            val = struct.pack("<I", data_off)
        out += struct.pack("<HHI4s", tag, typ, cnt, val)
    out += struct.pack("<I", 0) + extra + body
    path.write_bytes(out)


def _file_depth(path: pathlib.Path, head: bytes) -> int:
    """bits per sample of a PNG / TIFF file without decoding it (8 when unknown)"""
    try:
        if head[:8] == _PNG_SIG:
            return _png_header(head)[2]
        if head[:2] in (b"II", b"MM"):
            with open(path, "rb") as f:
                data = f.read()
            return max(_tiff_tags(data)[1].get(258, (8,)))
    except Exception:  # noqa: BLE001
        pass
    return 8


def to_uint8(arr: np.ndarray) -> np.ndarray:
    """16 -> 8 bits for 8-bit-only containers (JPEG): round(v * 255 / 65535)."""
    if arr.dtype == np.uint8:
        return arr
    return ((arr.astype(np.uint32) * 255 + 32767) // 65535).astype(np.uint8)


# ---- public --------------------------------------------------------------------------------------
def read_image(path) -> np.ndarray:
    """-> H x W x C (C = 1, 3 or 4; RGB order), uint8 -- or uint16 for 16-bit PNG / TIFF files (never silently reduced)."""
    path = pathlib.Path(path)
    ext = path.suffix.lower()
    try:
        with open(path, "rb") as f:
            head = f.read(64)
    except OSError as exc:
        raise ImageIOError(f"cannot read {path}: {exc}") from exc
    if ext in (".png", ".tif", ".tiff") and _file_depth(path, head) > 8:
        data = path.read_bytes()
        try:
            a = _png_read(data) if head[:8] == _PNG_SIG else _tiff_read16(data)
        except ImageIOError as exc:
            raise ImageIOError(f"cannot read {path}: {exc}") from exc
        except Exception as exc:  # noqa: BLE001
            raise ImageIOError(f"cannot read {path}: {exc}") from exc
        return np.ascontiguousarray(a)
    if Image is not None:
        try:
            with Image.open(path) as im:
                if im.mode in ("I;16", "I;16B", "I;16L", "I"):
                    raise ImageIOError("16-bit image in a container the 16-bit reader does not cover")
                if im.mode not in ("L", "RGB", "RGBA"):
                    im = im.convert("RGBA" if "A" in im.mode else "RGB")
                a = np.asarray(im)
        except ImageIOError as exc:
            raise ImageIOError(f"cannot read {path}: {exc}") from exc
        except Exception as exc:
            raise ImageIOError(f"cannot read {path}: {exc}") from exc
        return np.ascontiguousarray(a if a.ndim == 3 else a[:, :, None])
    if ext == ".png":
        return _png_read(path.read_bytes())
    raise ImageIOError(f"cannot read {path}: Pillow is not installed and the built-in codec is PNG only")


# Optimised Huffman tables = the reference's `-huffman optimal` (PC:331-338): a second pass over every image.  GS360_JPEG_OPTIMIZE=0
# skips it (slightly larger files); measured on the MI355X host it buys ~6 % end to end on the PerspCut CLI (46-47 -> 48-51 frames/s)
# and nothing on the dual-fisheye CLI, so the reference's setting stays the default.
_JPEG_OPTIMIZE = os.environ.get("GS360_JPEG_OPTIMIZE", "1") not in ("0", "off", "no")


def write_image(path, arr: np.ndarray, jpeg_q: int = None) -> None:
    """Encode by extension.  jpeg_q is ffmpeg's -q:v (1 = best, 2 ~ 95 %), mapped onto Pillow qualities
    with 4:4:4 sampling and optimised Huffman tables like the reference's mjpeg flags (PC:331-338)."""
    path = pathlib.Path(path)
    ext = path.suffix.lower()
    a = np.ascontiguousarray(arr)
    if a.ndim == 2:
        a = a[:, :, None]
    path.parent.mkdir(parents=True, exist_ok=True)
    if a.dtype == np.uint16:
        if ext == ".png":
            _png_write(path, a)
            return
        if ext in (".tif", ".tiff"):
            _tiff_write16(path, a)
            return
        # JPEG is an 8-bit container.  PerspCut: ffmpeg converts to yuvj444p, i.e. scales the depth -- the same thing.  The
        # dual-fisheye tool hands the uint16 array to cv2.imwrite (DF:749, 1215, 1838), whose JPEG encoder SATURATES
        # (convertTo(CV_8U): everything above 255 becomes 255); writing round(v * 255 / 65535) instead is a deliberate
        # deviation, not parity.
        a = to_uint8(a)
    if Image is not None:
        mode = {1: "L", 3: "RGB", 4: "RGBA"}[a.shape[2]]
        im = Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a, mode)
        if ext in (".jpg", ".jpeg"):
            if mode == "RGBA":
                im = im.convert("RGB")
            quality = 95 if (jpeg_q is not None and jpeg_q >= 2) else 100
            im.save(path, "JPEG", quality=quality, subsampling=0, optimize=_JPEG_OPTIMIZE)
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
