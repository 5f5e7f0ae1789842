"""16-bit image files stay 16-bit (round-1 VERDICT weak #6: read_image used to squeeze them to 8 bits silently)."""
import struct
import zlib

import numpy as np
import pytest

from gs360 import imageio


def rand16(h, w, c, seed=0):
    return np.random.default_rng(seed).integers(0, 65536, (h, w, c), dtype=np.uint16)


@pytest.mark.parametrize("channels", [1, 3, 4])
@pytest.mark.parametrize("ext", [".png", ".tif"])
def test_roundtrip_16bit(tmp_path, channels, ext):
    a = rand16(37, 53, channels, seed=channels)
    imageio.write_image(tmp_path / ("a" + ext), a)
    b = imageio.read_image(tmp_path / ("a" + ext))
    assert b.dtype == np.uint16 and np.array_equal(a, b)
    a8 = np.random.default_rng(9).integers(0, 256, (21, 17, channels), dtype=np.uint8)
    imageio.write_image(tmp_path / ("b" + ext), a8)
    b8 = imageio.read_image(tmp_path / ("b" + ext))
    assert b8.dtype == np.uint8 and np.array_equal(a8, b8)


def _png_with_all_filters(arr):
    """PNG encoder of the test: row y uses filter type y % 5 (None, Sub, Up, Average, Paeth)"""
    h, w, ch = arr.shape
    bpp = ch * 2
    rows = arr.astype(">u2").view(np.uint8).reshape(h, w * bpp).astype(np.int32)
    out = bytearray()
    prev = np.zeros(w * bpp, np.int32)
    for y in range(h):
        cur, ft = rows[y], y % 5
        line = np.zeros_like(cur)
        for i in range(len(cur)):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            line[i] = (cur[i] - pred) & 255
        out += bytes([ft]) + line.astype(np.uint8).tobytes()
        prev = cur

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, ctype, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(bytes(out))) + chunk(b"IEND", b"")


@pytest.mark.parametrize("channels", [1, 3, 4])
def test_png16_every_filter_type_c_helper_and_fallback(tmp_path, channels, monkeypatch):
    a = rand16(11, 9, channels, seed=5)
    (tmp_path / "f.png").write_bytes(_png_with_all_filters(a))
    assert np.array_equal(imageio.read_image(tmp_path / "f.png"), a)          # library helper (gs360_png_unfilter) when built
    from gs360 import capi
    monkeypatch.setattr(capi, "load_library", lambda *a_, **k: (_ for _ in ()).throw(RuntimeError("no library")))
    assert np.array_equal(imageio.read_image(tmp_path / "f.png"), a)          # pure-NumPy fallback


def test_16bit_files_are_never_reduced_silently(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    g = np.random.default_rng(1).integers(0, 65536, (40, 50), dtype=np.uint16)
    Image.fromarray(g).save(tmp_path / "g.png")                                # Pillow's adaptive filters
    assert np.array_equal(imageio.read_image(tmp_path / "g.png")[:, :, 0], g)
    Image.fromarray(g).save(tmp_path / "d.tif", compression="tiff_adobe_deflate")
    assert np.array_equal(imageio.read_image(tmp_path / "d.tif")[:, :, 0], g)
    Image.fromarray(g).save(tmp_path / "l.tif", compression="tiff_lzw")          # LZW through the library's host helper
    assert np.array_equal(imageio.read_image(tmp_path / "l.tif")[:, :, 0], g)
    runs = np.zeros((130, 517), np.uint16)
    runs[40:50] = 777                                                              # long runs: the KwKwK case of LZW
    Image.fromarray(runs).save(tmp_path / "r.tif", compression="tiff_lzw")
    assert np.array_equal(imageio.read_image(tmp_path / "r.tif")[:, :, 0], runs)
    Image.fromarray(g).save(tmp_path / "p.tif", compression="packbits")
    with pytest.raises(imageio.ImageIOError) as e:
        imageio.read_image(tmp_path / "p.tif")
    assert "not supported" in str(e.value)
    # JPEG is an 8-bit container: 16-bit data is scaled, not truncated
    rgb = rand16(16, 16, 3)
    assert np.array_equal(imageio.to_uint8(np.array([[[0, 32768, 65535]]], np.uint16))[0, 0], [0, 128, 255])
    imageio.write_image(tmp_path / "j.jpg", rgb, jpeg_q=1)
    assert imageio.read_image(tmp_path / "j.jpg").dtype == np.uint8


def _craft_tiff(arr, big_endian, predictor, rows_per_strip, deflate):
    """baseline TIFF writer of the test: chunky 16-bit, several strips, optional horizontal predictor / Deflate, either byte order"""
    h, w, ch = arr.shape
    e = ">" if big_endian else "<"
    strips = []
    for y in range(0, h, rows_per_strip):
        a = arr[y:y + rows_per_strip].astype(np.uint32)
        if predictor == 2:
            d = a.copy()
            d[:, 1:] = (a[:, 1:] - a[:, :-1]) & 0xFFFF
            a = d
        raw = a.astype(e + "u2").tobytes()
        strips.append(zlib.compress(raw) if deflate else raw)
    n_s = len(strips)
    entries = [(256, 4, [w]), (257, 4, [h]), (258, 3, [16] * ch), (259, 3, [8 if deflate else 1]), (262, 3, [2 if ch >= 3 else 1]),
               (273, 4, None), (277, 3, [ch]), (278, 4, [rows_per_strip]), (279, 4, [len(s_) for s_ in strips]), (284, 3, [1]),
               (317, 3, [predictor])]
    if ch == 4:
        entries.append((338, 3, [2]))
    entries.sort()
    ifd_off = 8
    extra_off = ifd_off + 2 + 12 * len(entries) + 4
    # strip offsets are known once the extra area is sized: lay out extras first with placeholder offsets
    fmt = {3: "H", 4: "I"}

    def pack(vals, typ):
        return struct.pack(e + fmt[typ] * len(vals), *vals)
    sizes = sum(len(pack(v if v is not None else [0] * n_s, t)) for _tag, t, v in entries if len(pack(v if v is not None else [0] * n_s, t)) > 4)
    data_off = extra_off + sizes + (sizes & 1)
    offs, pos = [], data_off
    for s_ in strips:
        offs.append(pos)
        pos += len(s_)
    extra, recs = b"", []
    for tag, typ, vals in entries:
        vals = offs if vals is None else vals
        blob = pack(vals, typ)
        if len(blob) <= 4:
            recs.append(struct.pack(e + "HHI", tag, typ, len(vals)) + blob.ljust(4, b"\0"))
        else:
            recs.append(struct.pack(e + "HHII", tag, typ, len(vals), extra_off + len(extra)))
            extra += blob
    extra += b"\0" * (data_off - extra_off - len(extra))
    return (b"MM" if big_endian else b"II") + struct.pack(e + "HI", 42, ifd_off) + struct.pack(e + "H", len(entries)) + \
        b"".join(recs) + struct.pack(e + "I", 0) + extra + b"".join(strips)


@pytest.mark.parametrize("big_endian,predictor,deflate", [(True, 1, False), (False, 2, False), (True, 2, True), (False, 1, True)])
@pytest.mark.parametrize("channels", [1, 3, 4])
def test_tiff16_byte_orders_predictor_strips(tmp_path, big_endian, predictor, deflate, channels):
    a = rand16(23, 31, channels, seed=7)
    (tmp_path / "t.tif").write_bytes(_craft_tiff(a, big_endian, predictor, 5, deflate))
    b = imageio.read_image(tmp_path / "t.tif")
    assert b.dtype == np.uint16 and np.array_equal(a, b)
