"""Video inputs: one shared decode, HBM-resident frames (gs360/video.py).

CPU: the decoder command derived from the planner's / the GUI's job argv, and the PPM stream reader.
GPU: the whole path with a test double standing in for the ffmpeg binary (tests/fake_ffmpeg.py): clip -> PPM pipe ->
device frames -> HIP views -> numbered files, compared with the oracle frame by frame."""
import argparse
import io
import os
import pathlib
import stat
import sys

import numpy as np
import pytest

from conftest import ROOT
from gs360 import imageio, planner, video
from gs360.jobspec import parse_job_argv
from util import HFOV_12MM

FAKE = ROOT / "tests" / "fake_ffmpeg.py"


def plan_jobs(tmp_path, extra, ffmpeg="ffmpeg", name="clip.mp4"):
    import gs360_360PerspCut as cut
    args = cut.create_arg_parser().parse_args(["-i", str(tmp_path / name), "--ffmpeg", ffmpeg] + extra)
    for attr in ("size", "hfov", "focal_mm"):
        setattr(args, f"{attr}_explicit", getattr(args, f"{attr}_explicit", False))
    args.input_is_video, args.video_bit_depth = True, 8
    return cut.build_view_jobs(args, [tmp_path / name], tmp_path / "out")


def gui_select_rewrite(cmd, indices):
    """what gs360_GUI.py:19081-19148 does to a planned argv for a CSV frame selection (restated for the test)"""
    cmd = list(cmd)
    seek = []
    for flag in ("-ss", "-to"):
        while flag in cmd:
            i = cmd.index(flag)
            cmd.pop(i)
            seek += [flag, cmd.pop(i)]
    v = cmd.index("-vf") + 1
    parts = [p for p in cmd[v].split(",") if not p.strip().startswith("fps=")]
    cmd[v] = ",".join(["select='" + "+".join(f"eq(n\\,{i})" for i in indices) + "'"] + parts)
    cmd[-1:-1] = ["-frame_pts", "1"]
    while "-start_number" in cmd:
        i = cmd.index("-start_number")
        del cmd[i:i + 2]
    cmd.insert(1, "-copyts")
    i = cmd.index("-i") + 2
    cmd[i:i] = seek
    return cmd


def test_gui_rewrite_helper_and_decode_plans_against_the_reference_gui_goldens():
    """tests/golden/gui_select_goldens.json holds argv lists rewritten by the reference GUI's own function
    (gs360_GUI.py:19081-19148): the helper above must reproduce them exactly, selections without a seek window get a shared
    decode plan numbered by the selected indices, selections WITH -ss/-to fall back (video.py:100)."""
    import json
    from conftest import GOLDEN
    cases = json.loads((GOLDEN / "gui_select_goldens.json").read_text())["cases"]
    assert len(cases) >= 5
    for name, g in cases.items():
        keys = set()
        for planned, rewritten in zip(g["planned"], g["rewritten"]):
            assert gui_select_rewrite(planned, g["indices"]) == rewritten, name
            job = parse_job_argv(rewritten)
            plan = video.build_decode_plan(job)
            if "-ss" in job.options or "-to" in job.options:
                assert plan is None, name
                continue
            assert plan is not None and plan.numbers == tuple(sorted(set(g["indices"]))), name
            assert "-copyts" in plan.argv and plan.argv[-1] == "pipe:1"
            assert [pathlib.Path(video.output_path(job, plan, k)).name for k in range(len(plan.numbers))] == \
                [job.dst.name % n for n in plan.numbers]
            keys.add(plan.key)
        assert len(keys) <= 1, name                       # every view job of the video shares ONE decode


def test_decode_plan_from_planner_argv(tmp_path):
    res = plan_jobs(tmp_path, ["-f", "2", "--ext", "png", "--start", "3", "--end", "12.5", "--count", "4"])
    plans = [video.build_decode_plan(parse_job_argv(cmd)) for cmd, _s, _d in res.jobs]
    assert all(p is not None for p in plans) and len({p.key for p in plans}) == 1      # every view shares one decode
    p = plans[0]
    src = str(tmp_path / "clip.mp4")
    assert list(p.argv) == ["ffmpeg", "-hide_banner", "-loglevel", "error", "-nostdin", "-ss", "3.0", "-i", src, "-to", "12.5",
                            "-vsync", "vfr", "-vf", "fps=2.0,colorspace=iall=bt709:all=smpte170m:trc=iec61966-2-1:format=yuv444p,format=rgb24",
                            "-an", "-f", "image2pipe", "-c:v", "ppm", "pipe:1"]
    assert p.numbers is None and p.start_number == 0
    job = parse_job_argv(res.jobs[1][0])
    assert video.output_path(job, p, 0).endswith("clip_0000000_B.png") and video.output_path(job, p, 41).endswith("clip_0000041_B.png")


def test_decode_plan_jpeg_and_gui_selection(tmp_path):
    res = plan_jobs(tmp_path, ["-f", "1", "--count", "2"])                    # jpg output: mjpeg/yuvj444p encoder options
    cmd = res.jobs[0][0]
    p = video.build_decode_plan(parse_job_argv(cmd))
    assert p is not None and "range=jpeg:format=yuv444p,format=rgb24" in p.argv[p.argv.index("-vf") + 1]
    assert "-q:v" not in p.argv and "-c:v" in p.argv and p.argv[p.argv.index("-c:v") + 1] == "ppm"
    sel = gui_select_rewrite(cmd, [40, 7, 19])
    job = parse_job_argv(sel)
    q = video.build_decode_plan(job)
    assert q is not None and q.numbers == (7, 19, 40) and q.key != p.key
    assert q.argv[5] == "-copyts" and "fps=" not in q.argv[q.argv.index("-vf") + 1]
    assert q.argv[q.argv.index("-vf") + 1].startswith("select='eq(n\\,40)+eq(n\\,7)+eq(n\\,19)',colorspace=")
    assert [pathlib.Path(video.output_path(job, q, k)).name for k in range(3)] == ["clip_0000007_A.jpg", "clip_0000019_A.jpg", "clip_0000040_A.jpg"]


def test_decode_plan_gui_selection_with_seek_falls_back(tmp_path):
    """ADVICE r1 (video.py:100): with -copyts and output-side -ss/-to ffmpeg drops selected frames outside [S, T]; the PPM
    pipe carries no timestamps, so the k-th decoded frame cannot be paired with the k-th selected index -> no plan"""
    cmd = plan_jobs(tmp_path, ["-f", "1", "--ext", "png", "--count", "2", "--start", "3", "--end", "9"]).jobs[0][0]
    assert video.build_decode_plan(parse_job_argv(cmd)) is not None                  # the planner's own shape is fine
    sel = gui_select_rewrite(cmd, [1, 4, 7, 20])
    assert "-ss" in sel and sel.index("-ss") > sel.index("-i")                       # the GUI moved the seek behind the input
    assert video.build_decode_plan(parse_job_argv(sel)) is None


def test_decode_plan_refuses_what_it_does_not_understand(tmp_path):
    cmd = plan_jobs(tmp_path, ["-f", "1", "--ext", "png", "--count", "2"]).jobs[0][0]
    assert video.build_decode_plan(parse_job_argv(cmd)) is not None

    def mutate(fn):
        c = list(cmd)
        fn(c)
        return video.build_decode_plan(parse_job_argv(c))
    deep = mutate(lambda c: c.__setitem__(c.index("-pix_fmt") + 1, "rgb48le"))                  # > 8-bit video (PC:343-347)
    assert deep is not None and deep.argv[deep.argv.index("-vf") + 1].endswith(",format=rgb48be")   # 16-bit PPM pipe
    assert mutate(lambda c: c.__setitem__(c.index("-pix_fmt") + 1, "gbrp12le")) is None         # any other pixel format
    assert mutate(lambda c: c.__setitem__(c.index("-vf") + 1, c[c.index("-vf") + 1] + ",hflip")) is None   # filter after v360
    assert mutate(lambda c: c.__setitem__(slice(-1, -1), ["-metadata", "x=y"])) is None          # foreign option
    assert mutate(lambda c: c.__setitem__(slice(-1, -1), ["-frame_pts", "1"])) is None           # pts numbering without a select list
    assert mutate(lambda c: c.__setitem__(c.index("-vf") + 1, "select='gt(scene\\,0.4)'," + c[c.index("-vf") + 1])) is None
    still = ["ffmpeg", "-y", "-i", "a.png", "-vf", "v360=input=equirect:output=rectilinear:w=8:h=8:yaw=0:pitch=0:roll=0:h_fov=90:v_fov=90:interp=cubic", "o.png"]
    assert video.build_decode_plan(parse_job_argv(still)) is None


def test_ppm_stream_reader():
    fr = np.arange(2 * 3 * 3, dtype=np.uint8)
    data = b"P6\n# made by a test\n3 2\n255\n" + fr.tobytes() + b"P6 3 2 255\n" + fr[::-1].tobytes()
    s = io.BufferedReader(io.BytesIO(data))
    for want in (fr, fr[::-1]):
        assert video.read_ppm_header(s) == (3, 2, 255)
        buf = bytearray(18)
        video.read_exact_into(s, memoryview(buf))
        assert bytes(buf) == want.tobytes()
    assert video.read_ppm_header(s) is None
    with pytest.raises(video.PpmError):
        video.read_ppm_header(io.BufferedReader(io.BytesIO(b"P5\n3 2\n255\n")))
    with pytest.raises(video.PpmError):
        video.read_ppm_header(io.BufferedReader(io.BytesIO(b"P6\n3 ")))
    with pytest.raises(video.PpmError):
        video.read_exact_into(io.BufferedReader(io.BytesIO(b"abc")), memoryview(bytearray(5)))


# ---- GPU, with the decoder test double ---------------------------------------------------------------------------
def fake_ffmpeg_program(tmp_path):
    p = tmp_path / "ffmpeg_double"
    p.write_text("#!/bin/sh\nexec {} {} \"$@\"\n".format(sys.executable, FAKE))
    p.chmod(p.stat().st_mode | stat.S_IXUSR)
    return str(p)


def make_clip(path, n=6, h=64, w=128):
    rng = np.random.default_rng(31)
    clip = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    np.save(path, clip)
    return clip


@pytest.mark.gpu
def test_video_views_through_shared_decode(tmp_path, orc):
    import gs360_360PerspCut as cut
    from gs360 import engine
    clip = make_clip(tmp_path / "clip.npy")
    res = plan_jobs(tmp_path, ["-f", "1", "--ext", "png", "--count", "3", "--size", "40", "--start", "1"],
                    ffmpeg=fake_ffmpeg_program(tmp_path), name="clip.npy")
    (tmp_path / "out").mkdir()
    cut.stop_event.clear()
    os.environ["GS360_INTERP"] = "linear"
    try:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=3) as pool:
            results = list(pool.map(cut.run_one, [cmd for cmd, _s, _d in res.jobs]))
    finally:
        os.environ.pop("GS360_INTERP")
    assert results == [(0, "")] * 3, results
    eng = engine.get_engine()
    assert not eng.videos                                   # all planned view jobs done -> frames released
    for k, (y, tag) in enumerate(((0.0, "A"), (120.0, "B"), (-120.0, "C"))):
        for n in range(5):                                  # frames 1..5 of the clip, numbered from 0
            got = imageio.read_image(tmp_path / "out" / f"clip_{n:07d}_{tag}.png")
            want = orc.equirect_views_u8(clip[n + 1], [orc.make_view(y, 0.0, HFOV_12MM, HFOV_12MM, 40, 40)])[0]
            assert np.array_equal(got, want), (tag, n)
    assert not (tmp_path / "out" / "clip_0000005_A.png").exists()


@pytest.mark.gpu
def test_full360coverage_video_walks_windows_through_the_source_major_kernel(tmp_path, orc, monkeypatch):
    """BASELINE configs[2] as the drop-in tool runs it (PC:746-749 video mode, PC:1049-1078 one worker per view job): the twelve view jobs of
    a `full360coverage` video walk the resident frames a WINDOW at a time, so that the library sees four frames per call and takes the
    source-major kernel for the ring family (one frame per call -- rounds 4-5 -- left it on the LDS-staged kernel); every written file
    equals the oracle"""
    import gs360_360PerspCut as cut
    from gs360 import engine, video
    from util import PRESET_FULL360
    # a window waits for its frames while the decoder is still delivering -- 50 ms in the product (a slower decoder is the bottleneck whatever
    # the window); the decoder double on a busy box can be slower than that, and this test is about FULL windows: wait until they are
    monkeypatch.setattr(video, "_WINDOW_WAIT_S", 20.0)
    monkeypatch.setattr(engine, "_LINGER_S", 5.0)           # ... and the leader of a window waits for all twelve view jobs (3 ms in the product)
    n_frames = 15                                           # windows of 4 + 4 + 4 + 3; the first one may go out before every view job has joined
    rng = np.random.default_rng(77)
    clip = rng.integers(0, 256, (n_frames, 512, 1024, 3), dtype=np.uint8)
    np.save(tmp_path / "clip.npy", clip)
    res = plan_jobs(tmp_path, ["-f", "1", "--ext", "png", "--preset", "full360coverage", "--size", "128"],
                    ffmpeg=fake_ffmpeg_program(tmp_path), name="clip.npy")
    assert len(res.jobs) == 12
    (tmp_path / "out").mkdir()
    cut.stop_event.clear()
    engine.shutdown()                                       # a fresh engine: its statistics are this run's
    os.environ["GS360_INTERP"] = "linear"
    try:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=12) as pool:
            results = list(pool.map(cut.run_one, [cmd for cmd, _s, _d in res.jobs]))
    finally:
        os.environ.pop("GS360_INTERP")
    assert results == [(0, "")] * 12, results
    st = engine.get_engine().stats()
    assert st["frames"] == 12 * n_frames or st["frames"] >= n_frames      # every frame went out (views of a window share launches)
    assert st["launches"] < st["frames"], st                # windows: fewer launches than frames
    assert video._WINDOW >= 4 and st.get("launches_srcmajor", 0) >= 2, st      # the four-frame windows ran eq_srcmajor_kernel
    names = sorted(p.name for p in (tmp_path / "out").iterdir())
    assert len(names) == 12 * n_frames
    assert sorted((v.yaw_deg, v.pitch_deg) for v in res.view_specs) == sorted((float(y), float(p_)) for y, p_ in PRESET_FULL360)
    for v in res.view_specs:
        for n in (0, 3, 4, 7, 10, 14):
            got = imageio.read_image(tmp_path / "out" / f"clip_{n:07d}_{v.view_id}.png")
            want = orc.equirect_views_u8(clip[n], [orc.make_view(v.yaw_deg, v.pitch_deg, v.hfov_deg, v.vfov_deg, 128, 128)])[0]
            assert np.array_equal(got, want), (v.view_id, n)


@pytest.mark.gpu
def test_video_gui_selection_numbering_and_decoder_failure(tmp_path, orc):
    import gs360_360PerspCut as cut
    clip = make_clip(tmp_path / "clip.npy")
    prog = fake_ffmpeg_program(tmp_path)
    res = plan_jobs(tmp_path, ["-f", "1", "--ext", "png", "--count", "2", "--size", "32"], ffmpeg=prog, name="clip.npy")
    (tmp_path / "out").mkdir()
    cut.stop_event.clear()
    cmd = gui_select_rewrite(res.jobs[1][0], [4, 0, 2])
    assert cut.run_one(cmd) == (0, "")
    names = sorted(p.name for p in (tmp_path / "out").glob("*.png"))
    assert names == ["clip_0000000_B.png", "clip_0000002_B.png", "clip_0000004_B.png"]
    got = imageio.read_image(tmp_path / "out" / "clip_0000004_B.png")
    want = orc.equirect_views_u8(clip[4], [orc.make_view(180.0, 0.0, HFOV_12MM, HFOV_12MM, 32, 32)], interp=2)[0]
    assert np.array_equal(got, want)
    np.save(tmp_path / "broken.npy", clip[:1])
    bad = plan_jobs(tmp_path, ["-f", "1", "--ext", "png", "--count", "2"], ffmpeg=prog, name="broken.npy").jobs[0][0]
    rc, text = cut.run_one(bad)
    assert rc == 1 and "decoder exited with code 1" in text and "cannot open input" in text
    # a > 8-bit video (rgb48le, PC:343-347) stays in the engine: 16-bit PPM pipe -> uint16 frames -> 16-bit PNG views
    deep = list(res.jobs[0][0])
    deep[deep.index("-pix_fmt") + 1] = "rgb48le"
    deep[-1] = str(tmp_path / "out" / "deep_%07d_A.png")
    assert cut.run_one(deep) == (0, "")
    got = imageio.read_image(tmp_path / "out" / "deep_0000003_A.png")
    want = orc.equirect_views_u16(clip[3].astype(np.uint16) * 257, [orc.make_view(0.0, 0.0, HFOV_12MM, HFOV_12MM, 32, 32)], interp=2)[0]
    assert got.dtype == np.uint16 and np.array_equal(got, want)
    # a pixel format the engine does not know goes to the per-view subprocess, which the double refuses (rc 3)
    odd = list(res.jobs[0][0])
    odd[odd.index("-pix_fmt") + 1] = "gbrp12le"
    rc, text = cut.run_one(odd)
    assert rc == 3 and "only the PPM pipe decoder role" in text


# ---- streaming residency (host logic; the decoder double runs as a real child process, device memory is faked) ------------
class _FakeCtx:
    handle = 1

    def __init__(self):
        self.live = {}
        self.n = 0
        self.lock = __import__("threading").Lock()

    class _Pinned:
        def __init__(self, n):
            self.view = memoryview(bytearray(n))

        def free(self):
            pass

    def pinned(self, nbytes):
        return self._Pinned(nbytes)

    def alloc(self, nbytes):
        with self.lock:
            self.n += 1
            self.live[self.n] = None
            return self.n

    def upload(self, buf, host, slot=0, sync=True):
        self.live[buf] = np.array(host, copy=True)

    def event_record(self, slot, idx):
        pass

    def event_sync(self, slot, idx):
        pass

    def free(self, buf):
        with self.lock:
            del self.live[buf]


class _FakeState:
    def __init__(self):
        import threading
        self.ctx = _FakeCtx()
        self.upload_slot = 3
        self.upload_lock = threading.Lock()


def _decode_plan(tmp_path, clip_name="clip.npy"):
    argv = (sys.executable, str(FAKE), "-hide_banner", "-loglevel", "error", "-nostdin", "-i", str(tmp_path / clip_name),
            "-vf", "format=rgb24", "-an", "-f", "image2pipe", "-c:v", "ppm", "pipe:1")
    return video.DecodePlan(argv, (str(tmp_path / clip_name), argv[1:]), None, 0)


def test_session_streams_three_times_its_budget_and_late_joiners_fall_back(tmp_path):
    """12 frames of 24 KB through a 2-device session with 16 KB per device (32 KB = one frame and a bit): the reader must retire
    what both view jobs have passed and wait for them otherwise; every job still sees every frame, in order, intact; a job that
    arrives after the first frames are gone cannot join."""
    import threading
    import time
    clip = make_clip(tmp_path / "clip.npy", n=12, h=64, w=128)
    states = [_FakeState(), _FakeState()]
    sess = video.VideoSession(states, _decode_plan(tmp_path), budget=16 << 10)
    toks = [sess.join(), sess.join()]
    assert toks == [0, 1]
    seen = {0: [], 1: []}

    def walk(tok, delay):
        k = 0
        while True:
            fr = sess.frame(tok, k)
            if fr is None:
                break
            st, buf, h, w, dt = fr
            seen[tok].append(st.ctx.live[buf].reshape(h, w, 3).copy())
            time.sleep(delay)
            k += 1
        sess.leave(tok)
    threads = [threading.Thread(target=walk, args=(0, 0.0)), threading.Thread(target=walk, args=(1, 0.01))]
    for t in threads:
        t.start()
    time.sleep(0.15)
    probe = sess.join()                                       # once frames were retired a newcomer is turned away
    if probe is not None:
        assert sess.first == 0
        sess.leave(probe)                                     # (it would otherwise hold every frame back)
    for t in threads:
        t.join(20)
    assert sess.error is None and sess.finished
    for tok in (0, 1):
        assert len(seen[tok]) == 12 and all(np.array_equal(a, b) for a, b in zip(seen[tok], clip))
    assert sess.retired >= 9 and sess.peak_bytes <= 2 * (16 << 10) + 24576   # streamed: never more than the budget (+ the frame in flight)
    assert sess.join() is None                                # frames 0.. are gone: late joiners need their own decode
    sess.close()
    assert all(not st.ctx.live for st in states)              # every device frame was freed


def test_session_without_pressure_keeps_everything_for_late_joiners(tmp_path):
    clip = make_clip(tmp_path / "clip.npy", n=5, h=16, w=32)
    sess = video.VideoSession([_FakeState()], _decode_plan(tmp_path), budget=1 << 20)
    a = sess.join()
    k = 0
    while sess.frame(a, k) is not None:
        k += 1
    sess.leave(a)
    assert k == 5 and sess.retired == 0
    b = sess.join()                                           # a view job that starts after the first one finished
    assert b is not None
    st, buf, h, w, dt = sess.frame(b, 4)
    assert np.array_equal(st.ctx.live[buf].reshape(h, w, 3), clip[4])
    sess.leave(b)
    sess.close()


def test_closing_a_blocked_session_releases_the_reader(tmp_path):
    make_clip(tmp_path / "clip.npy", n=8, h=64, w=128)
    sess = video.VideoSession([_FakeState()], _decode_plan(tmp_path), budget=30 << 10)   # one frame fits, nobody consumes
    import time
    t0 = time.monotonic()
    while sess.count < 1 and time.monotonic() - t0 < 30:      # (the decoder double is a python process: its start-up time is the machine's)
        time.sleep(0.05)
    time.sleep(0.3)                                           # the reader now sits at its budget with the second frame
    assert not sess.finished and sess.count >= 1
    sess.close()
    assert sess.finished and not sess.thread.is_alive()


def _walk_all(sess, tok, out, pause=None):
    k = 0
    while True:
        if pause is not None:
            pause.wait(10)
        fr = sess.frame(tok, k)
        if fr is None:
            break
        out.append(k)
        k += 1
    sess.leave(tok)


def test_two_sessions_of_one_engine_share_one_budget(tmp_path):
    """a superseded session stays alive next to the one that replaces it (engine._video_session): together they hold at most the
    engine's budget (+ the frames in flight), not one budget each (round-3 ADVICE, video.py:178)"""
    import threading
    make_clip(tmp_path / "clip.npy", n=10, h=64, w=128)       # 24 KB frames
    states = [_FakeState()]
    shared = video.SharedBudget(60 << 10)                      # two frames and a bit for BOTH sessions together
    peak = [0]
    real_add = shared.add

    def add(n):
        real_add(n)
        peak[0] = max(peak[0], shared.used)
    shared.add = add
    a = video.VideoSession(states, _decode_plan(tmp_path), shared=shared)
    b = video.VideoSession(states, _decode_plan(tmp_path), shared=shared)
    seen = {0: [], 1: []}
    threads = [threading.Thread(target=_walk_all, args=(s, s.join(), seen[i])) for i, s in enumerate((a, b))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(30)
        assert not t.is_alive()
    assert seen[0] == list(range(10)) and seen[1] == list(range(10)) and a.error is None and b.error is None
    assert peak[0] <= (60 << 10) + 2 * 24576                  # one frame in flight per session above the shared budget at most
    a.close()
    b.close()
    assert shared.used == 0 and not states[0].ctx.live


def test_device_out_of_memory_is_back_pressure_not_a_decoder_error(tmp_path):
    """the device refuses an allocation (other tenants, a budget set above what is free): the reader retires a passed frame or waits
    for the view jobs and tries again, exactly as at its budget; the job still sees every frame"""
    import threading
    from gs360 import capi
    clip = make_clip(tmp_path / "clip.npy", n=8, h=64, w=128)
    st = _FakeState()
    real_alloc = st.ctx.alloc
    refused = [0]

    def alloc(nbytes):                                          # the "device" holds two frames
        if len(st.ctx.live) >= 2:
            refused[0] += 1
            raise capi.Gs360Error(-5, "out of device memory")
        return real_alloc(nbytes)
    st.ctx.alloc = alloc
    sess = video.VideoSession([st], _decode_plan(tmp_path), budget=1 << 30)     # the budget alone would admit everything
    got = []
    tok = sess.join()
    k = 0
    while True:
        fr = sess.frame(tok, k)
        if fr is None:
            break
        _st, buf, h, w, _dt = fr
        got.append(_st.ctx.live[buf].reshape(h, w, 3).copy())
        k += 1
    sess.leave(tok)
    assert sess.error is None and len(got) == 8 and all(np.array_equal(a, b) for a, b in zip(got, clip))
    assert refused[0] > 0 and sess.retired >= 6
    sess.close()


@pytest.mark.gpu
def test_video_streams_past_the_budget_on_the_gpu(tmp_path, orc, monkeypatch):
    """the whole path with a budget of ~two frames: 10 frames x 3 views with only TWO workers, so that the third view job
    starts after the first frames were retired and is served by a second decode -- every output still equals the oracle"""
    import gs360_360PerspCut as cut
    from gs360 import engine
    clip = make_clip(tmp_path / "clip.npy", n=10, h=64, w=128)
    monkeypatch.setattr(video, "_BUDGET_BYTES", 2 * 64 * 128 * 3 + 100)
    res = plan_jobs(tmp_path, ["-f", "1", "--ext", "png", "--count", "3", "--size", "40"], ffmpeg=fake_ffmpeg_program(tmp_path), name="clip.npy")
    (tmp_path / "out").mkdir()
    cut.stop_event.clear()
    monkeypatch.setenv("GS360_INTERP", "linear")
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=2) as pool:
        results = list(pool.map(cut.run_one, [cmd for cmd, _s, _d in res.jobs]))
    assert results == [(0, "")] * 3, results
    assert not engine.get_engine().videos
    for y, tag in ((0.0, "A"), (120.0, "B"), (-120.0, "C")):
        for n in range(10):
            got = imageio.read_image(tmp_path / "out" / f"clip_{n:07d}_{tag}.png")
            want = orc.equirect_views_u8(clip[n], [orc.make_view(y, 0.0, HFOV_12MM, HFOV_12MM, 40, 40)])[0]
            assert np.array_equal(got, want), (tag, n)
