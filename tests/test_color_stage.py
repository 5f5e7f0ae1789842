"""Input colour stage (.cube LUT + Rec.709->sRGB re-encode, DF:494-725).

CPU: the NumPy oracle and the product's host tables against vectors captured from the reference
(tests/golden/make_color_goldens.py).  GPU: the HIP kernel through the C ABI against the oracle, bit-exact uint8."""
import json

import numpy as np
import pytest

import gs360
from conftest import ROOT
from gs360 import color
from oracle import color_np

G = np.load(ROOT / "tests" / "golden" / "color_goldens.npz")
META = json.loads((ROOT / "tests" / "golden" / "color_goldens.json").read_text())
LUTS = ("mix17", "id5", "dom9", "id2")


def same_power_as_golden_host():
    """NumPy's float32 power is implementation-defined; exact sRGB golden parity needs the same routine."""
    x = G["power_probe_in"]
    return (np.array_equal(np.power(x, 1.0 / 0.45), G["power_probe_out_045"])
            and np.array_equal(np.power(x, 1.0 / 2.4), G["power_probe_out_24"]))


def lut_of(name):
    return G[f"lut_{name}_table"], G[f"lut_{name}_dmin"], G[f"lut_{name}_dmax"]


def golden_cases():
    for key in G.files:
        if key.startswith("out_"):
            _, lut, img, space = key.split("_")
            yield lut, img, space


def assert_levels_close(got, want, exact):
    if exact:
        assert np.array_equal(got, want)
    else:   # another power routine: at most one level, on a tiny fraction of samples
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        lim = 1 if got.dtype == np.uint8 else 40
        assert d.max() <= lim and (d > 0).mean() < 2e-3


# ---- oracle vs reference vectors -----------------------------------------------------------------------------
@pytest.mark.parametrize("lut,img,space", sorted(golden_cases()))
def test_oracle_pipeline_matches_reference_vectors(lut, img, space):
    table, dmin, dmax = lut_of(lut)
    got = color_np.color_pipeline(G[f"img_{img}"], table, dmin, dmax, space, red_index=2)   # reference = BGR order
    want = G[f"out_{lut}_{img}_{space}"]
    assert got.dtype == want.dtype and got.shape == want.shape
    assert_levels_close(got, want, exact=(space == "passthrough" or same_power_as_golden_host()))


@pytest.mark.parametrize("lut", LUTS)
def test_oracle_trilinear_matches_reference_vectors(lut):
    table, dmin, dmax = lut_of(lut)
    got = color_np.trilinear(G[f"tri_{lut}_in"], table, dmin, dmax)
    assert np.array_equal(got.view(np.uint32), G[f"tri_{lut}_out"].view(np.uint32))     # float32 bit-exact


def test_oracle_transfer_functions_match_reference_vectors():
    x = G["tf_in"]
    exact = same_power_as_golden_host()
    for name, fn in (("tf_rec709_to_linear", color_np.rec709_to_linear), ("tf_linear_to_srgb", color_np.linear_to_srgb),
                     ("tf_rec709_to_srgb", lambda v: color_np.linear_to_srgb(color_np.rec709_to_linear(v)))):
        got = fn(x)
        if exact:
            assert np.array_equal(got.view(np.uint32), G[name].view(np.uint32)), name
        else:
            assert np.allclose(got, G[name], rtol=3e-7, atol=1e-7), name
    assert np.array_equal(color_np.from_float01(G["q8_in"], np.uint8), G["q8_out"])
    assert np.array_equal(color_np.from_float01(G["q8_in"], np.uint16), G["q16_out"])
    assert np.array_equal(color_np.to_float01(np.arange(256, dtype=np.uint8)).view(np.uint32), G["f01_u8"].view(np.uint32))


# ---- product host side ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lut", LUTS)
def test_cube_loader_matches_reference(lut, tmp_path):
    p = tmp_path / f"{lut}.cube"
    p.write_bytes(G[f"lut_{lut}_text"].tobytes())
    got = color.load_cube_lut(p)
    table, dmin, dmax = lut_of(lut)
    assert got.size == META[f"lut_{lut}"]["size"]
    assert np.array_equal(got.table, table) and got.table.dtype == np.float32
    assert np.array_equal(got.domain_min, dmin) and np.array_equal(got.domain_max, dmax)


@pytest.mark.parametrize("case", sorted(META["loader_errors"]))
def test_cube_loader_errors_match_reference(case, tmp_path):
    p = tmp_path / "bad.cube"
    p.write_bytes(G[f"bad_{case}_text"].tobytes())
    exc_name, text = META["loader_errors"][case]
    with pytest.raises(Exception) as e:
        color.load_cube_lut(p)
    assert type(e.value).__name__ == exc_name
    assert str(e.value).replace(str(p), "<path>") == text
    with pytest.raises(FileNotFoundError):
        color.load_cube_lut(tmp_path / "absent.cube")


def test_output_space_names():
    for k, v in META["normalize"].items():
        assert color.normalize_lut_output_color_space(k) == v
    with pytest.raises(ValueError):
        color.normalize_lut_output_color_space("rec2020")


def emulate_plan(stage, image, red_index):
    """The kernel's arithmetic in NumPy: host tables at both ends, the oracle's trilinear in between."""
    order = [0, 1, 2] if red_index == 0 else [2, 1, 0]
    lv = image[..., :3][..., order].reshape(-1, 3)
    n1 = stage.lut.size - 1
    pos = np.stack([stage.level_pos[c][lv[:, c]] for c in range(3)], -1)
    # feed positions back through the oracle's interpolation by inverting pos -> coordinate on an identity domain
    i0 = np.floor(pos).astype(np.int32)
    i1 = np.minimum(i0 + 1, n1)
    t = pos - i0.astype(np.float32)
    T = stage.lut.table

    def lerp(a, b, w):
        return a + (b - a) * w
    acc = []
    for bi in (i0[:, 2], i1[:, 2]):
        rows = [lerp(T[bi, gi, i0[:, 0]], T[bi, gi, i1[:, 0]], t[:, 0:1]) for gi in (i0[:, 1], i1[:, 1])]
        acc.append(lerp(rows[0], rows[1], t[:, 1:2]))
    x = lerp(acc[0], acc[1], t[:, 2:3])
    q = np.searchsorted(stage.thresholds[1:], np.clip(x, 0.0, 1.0), side="right").astype(np.uint8).reshape(image.shape[:2] + (3,))[..., order]
    out = image.copy()
    out[..., :3] = q
    return out


@pytest.mark.parametrize("lut,img,space", sorted(c for c in golden_cases() if c[1] != "u16"))
def test_host_tables_reproduce_reference_vectors(lut, img, space):
    """level positions + thresholds (what the kernel consumes) give the reference's bytes"""
    table, dmin, dmax = lut_of(lut)
    stage = color.ColorStage(color.CubeLUT(table.shape[0], table, dmin, dmax), space)
    got = emulate_plan(stage, G[f"img_{img}"], red_index=2)
    assert_levels_close(got, G[f"out_{lut}_{img}_{space}"], exact=(space == "passthrough" or same_power_as_golden_host()))


@pytest.mark.parametrize("space", ["srgb", "passthrough"])
def test_thresholds_are_the_exact_step_positions(space):
    thr = color.output_thresholds(space)
    assert thr.shape == (256,) and thr.dtype == np.float32 and np.all(np.diff(thr[1:]) >= 0)
    fin = thr[1:][np.isfinite(thr[1:])]
    k = np.arange(1, 1 + fin.size)
    assert np.array_equal(color.encode_levels(fin, space), k)                         # at the threshold: level k
    below = np.nextafter(fin, np.float32(-1))
    assert np.all(color.encode_levels(below[1:], space) < k[1:])                       # one float below: still k-1
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(200000, dtype=np.float32), np.float32([-1.0, 0.0, 1.0, 2.0])])
    got = np.searchsorted(thr[1:], x, side="right")
    want = color.encode_levels(x, space).astype(np.int64)
    ok = ~np.isnan(x)
    assert np.array_equal(got[ok], want[ok])
    if space == "passthrough":
        assert thr[1] == np.float32(0.5 / 255.0 + 1e-12) or color.encode_levels(thr[1:2], space)[0] == 1


def test_stage_rejects_what_the_reference_rejects():
    with pytest.raises(ValueError, match="at least 3-channel"):
        color.ColorStage.check_image((4, 4), np.uint8)
    with pytest.raises(ValueError, match="at least 3-channel"):
        color.ColorStage.check_image((4, 4, 1), np.uint8)
    color.ColorStage.check_image((4, 4, 3), np.uint16)            # 16-bit images are handled since round 2
    with pytest.raises(TypeError):
        color.ColorStage.check_image((4, 4, 3), np.float32)


@pytest.mark.parametrize("space", ["srgb", "passthrough"])
def test_output_pieces16_reproduce_the_encode_step(space):
    """the 16-bit quantiser data handed to the kernel: piece selection + threshold count == NumPy's own encode, including
    the non-monotone step across the Rec.709 knee (piece 2 starts BELOW the level piece 1 ends on)"""
    n, start, base, off, thr = color.output_pieces16(space)
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.random(300000, dtype=np.float32), np.float32([-1.0, 0.0, 1.0, 2.0, 0.081, 0.0140886]),
                        np.nextafter(np.float32(0.081), np.float32(0)).reshape(1)])
    xc = np.clip(x, 0, 1)
    if n == 0:
        assert thr.size == 0
        got = np.rint(xc * np.float32(65535.0)).astype(np.int64)
    else:
        assert n == 3 and off[3] == thr.size and base[2] < base[1] + (off[2] - off[1])
        piece = (xc >= start[1]).astype(int) + (xc >= start[2]).astype(int)
        got = np.empty(x.size, np.int64)
        for q in range(3):
            m = piece == q
            got[m] = base[q] + np.searchsorted(thr[off[q]:off[q + 1]], xc[m], side="right")
    assert np.array_equal(got, color.encode_levels16(x, space).astype(np.int64))


# ---- GPU parity --------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    c = gs360.Context(device=0, n_slots=2)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lut,img,space", sorted(c for c in golden_cases() if c[1] != "u16"))
def test_gpu_matches_reference_vectors_and_oracle(ctx, lut, img, space):
    table, dmin, dmax = lut_of(lut)
    stage = color.ColorStage(color.CubeLUT(table.shape[0], table, dmin, dmax), space)
    image = G[f"img_{img}"]
    for red in (2, 0):
        got = stage.apply(ctx, image, red_index=red)
        want = color_np.color_pipeline(image, table, dmin, dmax, space, red_index=red)
        assert np.array_equal(got, want), (lut, img, space, red)
        if red == 2:
            assert_levels_close(got, G[f"out_{lut}_{img}_{space}"],
                                exact=(space == "passthrough" or same_power_as_golden_host()))
    stage.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(37, 101, 3), (64, 256, 3), (5, 1023, 4), (3, 1, 3), (1, 7, 4), (480, 1920, 3)])
def test_gpu_shapes_and_alignment(ctx, shape):
    """odd widths take the byte path, dword-aligned rows the 4-pixel path; both must equal the oracle"""
    table, dmin, dmax = lut_of("dom9")
    stage = color.ColorStage(color.CubeLUT(9, table, dmin, dmax), "srgb")
    image = np.random.default_rng(shape[1]).integers(0, 256, shape, dtype=np.uint8)
    got = stage.apply(ctx, image, red_index=0)
    assert np.array_equal(got, color_np.color_pipeline(image, table, dmin, dmax, "srgb", red_index=0))
    stage.close()


@pytest.mark.gpu
def test_gpu_full_size_lens_image_33_cube(ctx):
    """a 4000x4000 lens image through a 33^3 LUT (the common .cube size): oracle parity on a strided sample of rows,
    plus the alpha-free in-place path on device"""
    n = 33
    g = np.linspace(0, 1, n, dtype=np.float32)
    bb, gg, rr = np.meshgrid(g, g, g, indexing="ij")
    table = np.stack([rr ** 0.8, 0.9 * gg + 0.1 * bb, np.sqrt(bb)], -1).astype(np.float32)
    dmin, dmax = np.zeros(3, np.float32), np.ones(3, np.float32)
    stage = color.ColorStage(color.CubeLUT(n, table, dmin, dmax), "srgb")
    rng = np.random.default_rng(11)
    base = rng.integers(0, 256, (250, 4000, 3), dtype=np.uint8)
    image = np.tile(base, (16, 1, 1))
    image[::7] = np.roll(image[::7], 3, axis=1)
    got = stage.apply(ctx, image, red_index=0)
    rows = np.r_[0:4000:97, 3999]
    want = color_np.color_pipeline(image[rows], table, dmin, dmax, "srgb", red_index=0)
    assert np.array_equal(got[rows], want)
    stage.close()


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["srgb", "passthrough"])
def test_gpu_cube_equals_per_pixel_evaluation_on_every_colour(ctx, space):
    """the 2^24-entry cube a plan applies is the per-pixel evaluation of every 8-bit colour: one 4096x4096 image that holds each
    colour once goes through a plan with the cube and through one without (context option "color_cube" = 0), RGB and BGR, and a strided sample
    of rows through the oracle"""
    table, dmin, dmax = lut_of("dom9")
    v = np.arange(1 << 24, dtype=np.uint32)
    rng = np.random.default_rng(24)
    rng.shuffle(v)
    image = np.stack([v & 255, (v >> 8) & 255, v >> 16], -1).astype(np.uint8).reshape(4096, 4096, 3)
    rows = np.r_[0:4096:173, 4095]
    for red in (0, 2):
        out = {}
        for cube in ("1", "0"):
            with ctx.options(color_cube=int(cube)):
                stage = color.ColorStage(color.CubeLUT(table.shape[0], table, dmin, dmax), space)
                out[cube] = stage.apply(ctx, image, red_index=red)
                stage.close()
        assert np.array_equal(out["1"], out["0"]), (space, red)
        want = color_np.color_pipeline(image[rows], table, dmin, dmax, space, red_index=red)
        assert np.array_equal(out["1"][rows], want), (space, red)
    # 4 channels (alpha passes through) and the byte path (odd width) read the same cube
    ctx.set_option("color_cube", 1)
    stage = color.ColorStage(color.CubeLUT(table.shape[0], table, dmin, dmax), space)
    rgba = np.concatenate([image[:64, :1021], rng.integers(0, 256, (64, 1021, 1), dtype=np.uint8)], -1)
    got = stage.apply(ctx, rgba, red_index=2)
    assert np.array_equal(got[..., 3], rgba[..., 3])
    assert np.array_equal(got[..., :3], color_np.color_pipeline(np.ascontiguousarray(rgba[..., :3]), table, dmin, dmax, space, red_index=2))
    stage.close()
    ctx.set_option("color_cube", -1)


@pytest.mark.gpu
@pytest.mark.parametrize("name,thr_of_k", [("one_per_bin", lambda k: k / 520.0), ("two_per_bin", lambda k: 0.3 + k / 1800.0),
                                            ("dense", lambda k: 0.5 + k * 1e-5), ("ties_and_never", lambda k: np.where(k < 200, (k // 4) / 64.0, np.inf)),
                                            ("zero_start", lambda k: np.maximum(k - 3, 0) / 300.0)])
def test_gpu_output_quantiser_variants(ctx, name, thr_of_k):
    """the bin-table quantiser (1 or 2 compares) and the binary-search fallback count thresholds exactly"""
    table, dmin, dmax = lut_of("id2")                      # identity cube: LUT output = level / 255
    pos = color.level_positions(color.CubeLUT(2, table, dmin, dmax))
    thr = np.empty(256, np.float32)
    thr[0] = -np.inf
    thr[1:] = thr_of_k(np.arange(1, 256))
    plan = ctx.color_plan(table, pos, thr)
    img = np.random.default_rng(2).integers(0, 256, (33, 64, 3), dtype=np.uint8)
    img[0, :, 0] = np.arange(64) * 4
    img[1, :, 1] = 255 - np.arange(64)
    d = ctx.to_device(img)
    ctx.color_apply_dev(plan, d, 33, 64, 3)
    got = ctx.download(d, img.shape)
    x = color_np.trilinear(img.astype(np.float32) / np.float32(255.0), table, dmin, dmax)
    want = np.searchsorted(thr[1:], np.clip(x, 0, 1), side="right").astype(np.uint8)
    assert np.array_equal(got, want), name
    ctx.free(d)
    ctx.color_plan_free(plan)


@pytest.mark.gpu
def test_gpu_plan_argument_checks(ctx):
    table, dmin, dmax = lut_of("id5")
    pos = color.level_positions(color.CubeLUT(5, table, dmin, dmax))
    thr = color.output_thresholds("passthrough")
    bad_pos = pos.copy()
    bad_pos[1, 7] = 4.5
    with pytest.raises(gs360.Gs360Error):
        ctx.color_plan(table, bad_pos, thr)
    bad_thr = thr.copy()
    bad_thr[10] = 0.9
    with pytest.raises(gs360.Gs360Error):
        ctx.color_plan(table, pos, bad_thr)
    plan = ctx.color_plan(table, pos, thr)
    buf = ctx.alloc(64)
    with pytest.raises(gs360.Gs360Error):
        ctx.color_apply_dev(plan, buf, 2, 2, 1)
    ctx.free(buf)
    ctx.color_plan_free(plan)


# ---- 16-bit images on the GPU -------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("lut,img,space", sorted(c for c in golden_cases() if c[1] == "u16"))
def test_gpu_u16_matches_reference_vectors_and_oracle(ctx, lut, img, space):
    table, dmin, dmax = lut_of(lut)
    stage = color.ColorStage(color.CubeLUT(table.shape[0], table, dmin, dmax), space)
    image = G[f"img_{img}"]
    assert image.dtype == np.uint16
    for red in (2, 0):
        got = stage.apply(ctx, image, red_index=red)
        want = color_np.color_pipeline(image, table, dmin, dmax, space, red_index=red)
        assert got.dtype == np.uint16 and np.array_equal(got, want), (lut, space, red)
        if red == 2:      # the reference's own output (NumPy power differs between hosts by one ulp -> one 16-bit level)
            assert_levels_close(got, G[f"out_{lut}_{img}_{space}"], exact=(space == "passthrough" or same_power_as_golden_host()))
    stage.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,space", [((37, 101, 3), "srgb"), ((5, 1023, 4), "srgb"), ((64, 256, 3), "passthrough"), ((1, 7, 4), "passthrough"),
                                          ((400, 1600, 3), "srgb"), ((33, 102, 3), "srgb"), ((9, 2, 3), "srgb")])
def test_gpu_u16_shapes_alpha_and_full_range(ctx, shape, space):
    table, dmin, dmax = lut_of("dom9")
    stage = color.ColorStage(color.CubeLUT(9, table, dmin, dmax), space)
    image = np.random.default_rng(shape[1]).integers(0, 65536, shape, dtype=np.uint16)
    image[0, :, 0] = np.linspace(0, 65535, shape[1]).astype(np.uint16)        # sweeps every piece of the quantiser
    for red in (0, 2):    # dword-aligned rows take the two-pixels-per-thread kernel (odd widths: its one-pixel tail), others the 16-bit one
        got = stage.apply(ctx, image, red_index=red)
        assert np.array_equal(got, color_np.color_pipeline(image, table, dmin, dmax, space, red_index=red)), red
    stage.close()
