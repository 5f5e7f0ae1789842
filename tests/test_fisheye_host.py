"""Host-side dual-fisheye geometry (gs360/fisheye.py) against vectors captured from the reference's NumPy builders.
Bit-for-bit under NumPy >= 2 (the goldens were captured with NumPy 2.2.6; SURVEY section 7 explains the 1.x delta)."""
import json
import pathlib

import numpy as np
import pytest

from conftest import GOLDEN
from gs360 import fisheye as fe
from util import TEMPLATE_CALIB

G = np.load(GOLDEN / "df_goldens.npz")
M = json.loads((GOLDEN / "df_goldens.json").read_text())
NP2 = int(np.__version__.split(".")[0]) >= 2
pytestmark = pytest.mark.skipif(not NP2, reason="goldens captured under NumPy 2 promotion rules")


def calib(d):
    return fe.SensorCalibration(d["sensor_id"], d["model_type"], d["width"], d["height"],
                                *[float(d[k]) for k in ("f", "cx", "cy", "k1", "k2", "k3", "k4", "p1", "p2", "b1", "b2")])


TMPL, FULL = calib(M["template_calibration"]), calib(M["synthetic_calibration"])


def test_template_xml_roundtrip(tmp_path):
    """write a Metashape-style XML with the template's numbers and read it back (adjusted beats initial)"""
    d = M["template_calibration"]
    xml = f"""<document><chunk><sensors>
      <sensor id="0" label="unknown" type="equisolid_fisheye"><resolution width="{d['width']}" height="{d['height']}"/>
        <calibration type="equisolid_fisheye" class="initial"><resolution width="{d['width']}" height="{d['height']}"/><f>1050</f></calibration>
        <calibration type="equisolid_fisheye" class="adjusted"><resolution width="{d['width']}" height="{d['height']}"/>
          <f>{d['f']}</f><cx>{d['cx']}</cx><cy>{d['cy']}</cy><k1>{d['k1']}</k1><k2>{d['k2']}</k2><k3>{d['k3']}</k3></calibration>
      </sensor>
      <sensor id="7" type="frame"><resolution width="10" height="10"/><calibration class="adjusted"><f>0</f></calibration></sensor>
      <sensor id="8" type="equisolid_fisheye"><calibration class="adjusted"><resolution width="5" height="5"/><f>3</f></calibration></sensor>
    </sensors><cameras><camera id="0" sensor_id="0" label="a_X"/><camera id="1" sensor_id="0" label="a_Y"/><camera id="2" label="nolabel"/></cameras></chunk></document>"""
    p = tmp_path / "c.xml"
    p.write_text(xml)
    sensors, labels = fe.load_metashape_calibration(p)
    assert set(sensors) == {"0"}                     # f <= 0 dropped; sensor 8 has no sensor-level resolution node
    c = sensors["0"]
    for k in ("f", "cx", "cy", "k1", "k2", "k3", "k4", "p1", "p2", "b1", "b2"):
        assert repr(getattr(c, k)) == M["template_calibration"][k], k
    assert (c.width, c.height, c.model_type) == (3840, 3840, "equisolid_fisheye")
    assert labels == {"a_X": "0", "a_Y": "0"}


def test_sfm10_and_helpers():
    got = fe.sfm10_specs(1750, 14.0, "36 36", 40.0, 40.0)
    assert [{k: (repr(v) if isinstance(v, float) else v) for k, v in s.items()} for s in got] == M["sfm10_specs_default"]
    got = fe.sfm10_specs(512, 18.0, "36x24", 35.0, 25.0)
    assert [{k: (repr(v) if isinstance(v, float) else v) for k, v in s.items()} for s in got] == M["sfm10_specs_alt"]
    for f, s, want in M["compute_view_fov_deg"]:
        assert [repr(x) for x in fe.view_fov_deg(f, s)] == want
    for a, want in M["wrap_angle_deg"]:
        assert repr(fe.wrap_angle_deg(a)) == want
    for bad in [dict(output_size=0), dict(yaw_delta_deg=180.0), dict(pitch_delta_deg=89.9), dict(focal_mm=0.0),
                dict(sensor_mm="abc")]:
        kw = dict(output_size=10, focal_mm=14.0, sensor_mm="36 36", yaw_delta_deg=40.0, pitch_delta_deg=40.0)
        kw.update(bad)
        with pytest.raises(ValueError):
            fe.sfm10_specs(**kw)


def test_brown_and_rotation_bit_exact():
    for tag, c in (("tmpl", TMPL), ("full", FULL)):
        xd, yd, r2 = fe.brown_distort(G["brown_in_x"], G["brown_in_y"], c)
        assert xd.dtype == np.float32
        assert np.array_equal(xd, G[f"brown_{tag}_xd"]) and np.array_equal(yd, G[f"brown_{tag}_yd"])
        assert np.array_equal(r2, G[f"brown_{tag}_r2"])
    for i, (yw, pt) in enumerate(M["rot_cases"]):
        assert np.array_equal(fe.rotate_pitch_yaw(G["rot_in"], yw, pt), G[f"rot_out_{i}"])


@pytest.mark.parametrize("case", M["small_cases"], ids=[c[0] for c in M["small_cases"]])
def test_perspective_tables_bit_exact(case):
    name, cname, yaw, pitch, hf, vf, w, h, lf = case
    mx, my, valid = fe.perspective_tables(TMPL if cname == "tmpl" else FULL, yaw, pitch, hf, vf, w, h, lf)
    assert mx.dtype == np.float32 and valid.dtype == np.bool_
    assert np.array_equal(mx, G[name + "_mx"]) and np.array_equal(my, G[name + "_my"])
    assert np.array_equal(valid, G[name + "_valid"])


def test_perspective_tables_real_size_view():
    hf = float(M["sfm10_specs_default"][0]["hfov_deg"])
    st = M["real_stride"]
    vid, yaw, pitch = M["real_views"][1]      # A_U
    mx, my, valid = fe.perspective_tables(TMPL, yaw, pitch, hf, hf, 1750, 1750, 190.0)
    assert np.array_equal(mx[::st, ::st], G[f"real_{vid}_mx_s"]) and np.array_equal(my[::st, ::st], G[f"real_{vid}_my_s"])
    assert np.array_equal(valid[::st, ::st], G[f"real_{vid}_valid_s"])
    assert repr(float(np.mean(valid))) == M[f"real_{vid}_valid_ratio"]


def test_lens_choice_and_tie_break():
    specs = fe.sfm10_specs(175, 14.0, "36 36", 40.0, 40.0)
    sel = fe.choose_lens_tables({"0": TMPL}, "0", "0", specs, 0.0, 180.0, 190.0)
    assert {k: v["lens_key"] for k, v in sel.items()} == M["lens_choice_175"]
    assert {k: repr(float(np.mean(v["valid"]))) for k, v in sel.items()} == M["lens_valid_ratio_175"]
    for k, v in sel.items():
        assert np.array_equal(v["map_x"][::5, ::5], G[f"sel175_{k}_mx"]) and np.array_equal(v["map_y"][::5, ::5], G[f"sel175_{k}_my"])
    sel2 = fe.choose_lens_tables({"0": TMPL}, "0", "0", specs, 20.0, -150.0, 150.0)
    assert {k: v["lens_key"] for k, v in sel2.items()} == M["lens_choice_175_rig2"]
    assert {k: repr(float(np.mean(v["valid"]))) for k, v in sel2.items()} == M["lens_valid_ratio_175_rig2"]
    for k in ("B", "E", "A_U"):
        assert np.array_equal(sel2[k]["valid"], G[f"sel175rig2_{k}_valid"])


def test_undistort_tables_and_auto_zoom():
    t = fe.undistort_tables(FULL, 1.35, 170.0)
    assert np.array_equal(t.map_x, G["undist_full_mx"]) and np.array_equal(t.map_y, G["undist_full_my"])
    assert np.array_equal(t.valid_mask, G["undist_full_valid"]) and repr(t.undistort_zoom) == M["undistort_zoom_full_explicit"]
    for f_scale, lf, want in M["auto_zoom_cases_full"]:
        c = fe.SensorCalibration(**{**FULL.__dict__, "f": FULL.f * f_scale})
        assert repr(fe.auto_undistort_zoom(c, lens_fov_deg=lf)) == want
    c = fe.SensorCalibration(**{**FULL.__dict__, "f": FULL.f * 1.6})
    t = fe.undistort_tables(c, None, 100.0)
    assert repr(t.undistort_zoom) == M["undistort_zoom_full_auto_f1.6_lf100"]
    assert np.array_equal(t.map_x[::4, ::4], G["undist_full_auto_mx"]) and np.array_equal(t.valid_mask[::4, ::4], G["undist_full_auto_valid"])
    with pytest.raises(ValueError):
        fe.undistort_tables(fe.SensorCalibration("s", "frame", 8, 8, 10.0), 1.0, 190.0)


def test_lens_table_builds_are_the_same_on_the_thread_pool(monkeypatch):
    """choose_lens_tables runs its 2 x len(specs) independent builds on a thread pool: same tables, same lens choice, same order
    as one after the other (GS360_TABLE_BUILD_THREADS=1, the reference's loop DF:1857-1907)."""
    c = fe.SensorCalibration("0", "equisolid_fisheye", 480, 480, TEMPLATE_CALIB["f"] / 8, TEMPLATE_CALIB["cx"], TEMPLATE_CALIB["cy"],
                             TEMPLATE_CALIB["k1"], TEMPLATE_CALIB["k2"], TEMPLATE_CALIB["k3"])
    specs = fe.sfm10_specs(97, 14.0, "36 36", 40.0, 40.0)
    monkeypatch.setattr(fe, "_TABLE_BUILD_THREADS", 6)
    a = fe.choose_lens_tables({"0": c}, "0", "0", specs, 0.0, 180.0, 190.0)
    monkeypatch.setattr(fe, "_TABLE_BUILD_THREADS", 1)
    b = fe.choose_lens_tables({"0": c}, "0", "0", specs, 0.0, 180.0, 190.0)
    assert list(a) == list(b) == [s["view_id"] for s in specs]
    for k in a:
        assert a[k]["lens_key"] == b[k]["lens_key"] and a[k]["yaw_rel_deg"] == b[k]["yaw_rel_deg"]
        for f in ("map_x", "map_y", "valid"):
            assert a[k][f].tobytes() == b[k][f].tobytes(), (k, f)
