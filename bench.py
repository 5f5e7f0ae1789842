#!/usr/bin/env python3
"""bench.py -- headline benchmark of the 360PerspCut reprojection hot path on MI355X.

Metric (BASELINE.json): MPix/s remapped, 8K equirect -> preset views; % of HBM roofline.
Workload (BASELINE.json configs[1]): 7680x3840x3 uint8 equirect frames -> `--preset default --count 6
--size 800` (6 x 800^2 views, f=12 mm -> hfov=vfov=112.62 deg), uint8 fixed-point bilinear.

One "step" = ONE batched launch of gs360_equirect_views_u8 over `--frames` DISTINCT frames that are
already resident in HBM (default 16 frames = the C ABI's per-launch maximum = 1.4 GB of source, 5.5x the 256 MiB
Infinity Cache, so every step is served from HBM; measured: 1 frame/step (cache-hot) and 4 frames/step are ~8 % and
~18 % faster per frame and are NOT what is reported; 8 frames/step is 1-2 % slower per frame than 16 because a launch's
last partial round of workgroups weighs twice as much).  `value` = output pixels written by all ranks / wall time.

The timed job is FIXED: `--steps` steps of L launches of `--frames` frames; L = 1 on one rank (the headline run), 16 on N > 1
ranks (a rank's timed share at the driver's --steps 20 is then 12 ms at N = 8 instead of 0.8 ms; `value` is a rate, so the
lines compare across N).  The launches of that job are dealt round-robin to the ranks
(gs360/sharding.py, the partition the engine uses), every rank renders its share from its own HBM-resident frames, and `value` = the job's pixels / the slowest rank's time (common start after barrier + synchronize, each
rank's clock stops when its own work is synchronised, MAX over ranks; the closing barrier follows and the time including it is
reported as `config.seconds_incl_closing_barrier`): strong scaling of a resident job
(`"scaling": "strong"`; at N = 1 it is exactly the headline run).  Per-rank times travel in `config.per_rank_seconds`.
Before the W warm-up steps every rank runs the same launches untimed for `--settle-ms` (150 ms): the device's clocks ramp over
the first ~100 ms of load, and a 20-step run behind 5 warm-up steps (8 ms in all) otherwise measures the ramp, not the kernel.

    python bench.py --gpus 1 --steps 400 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...            # no launcher in the environment: starts that torch.distributed.run itself

Frames x views shard with no exchange step, so no collective touches the data path; torch.distributed is used only
for the barrier, the max-over-ranks of the elapsed time and the gather of the per-rank times.

`--mode job` is BASELINE.json configs[2] kernel-only: `--job-frames` (600) synthetic 8K frames resident in HBM (53 GB on one
GPU, 600/N per rank) -> `full360coverage` 12 x 1600^2, the state gs360/video.py leaves the devices in after the shared decode
of a video; frames dealt to the ranks the same way, strong scaling.

`--mode stream` is the host-fed STRONG-scaling companion (BASELINE.json configs[2], reference seam
gs360_360PerspCut.py:1049-1078): 600 8K frames are dealt round-robin to the ranks (gs360/sharding.py), every frame
goes pinned host -> H2D, four frames at a time through one 4 x 12-view `full360coverage` launch, -> D2H (gs360/stream.py); the rate is bounded by
PCIe (88.5 MB in + 92.2 MB out per frame), which the line reports next to the measured value.  It is never the
headline `value` of the default mode.
"""
import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "360cam-pgm-3dgs-tools_amd"))

W, H, C = 7680, 3840, 3
N_VIEWS, SIZE = 6, 800
HFOV = 112.61986494804043          # fov_from_focal_mm(12, 36)  (reference PC:77-78)
# Algorithmic bytes of one frame of this workload:
#   store 6*800*800*3 = 11,520,000 B  +  distinct source texels touched by the bilinear taps
#   sum_v U_v = 14,325,324 texels * 3 B = 42,975,972 B   (counted by the oracle; tests/test_oracle_equirect.py)
ALGO_BYTES_PER_FRAME = 11_520_000 + 14_325_324 * 3
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# The bound a gather that moves whole 128-B lines can reach (profiles/HISTORY.md): every view has to pull each distinct
# line its taps touch at least once -- 718,080 lines per frame summed over the six views (119,680 each; counted from the
# oracle's map, tests/test_oracle_equirect.py) -- plus the stores, at the 6.29 TB/s the HBM sustains for streaming copies.
LINE_BYTES_PER_FRAME = 718_080 * 128 + 11_520_000
# ... and the bound of the source-major kernel (round 5), which pulls each distinct line ONCE for all six views: the union of the views'
# line sets is 413,172 lines per frame (tests/test_oracle_equirect.py), + the stores
UNION_LINE_BYTES_PER_FRAME = 413_172 * 128 + 11_520_000
# SURVEY 8(d)'s STRICTER figure: the union over the six views of a frame, every distinct texel once (11,685,864 texels, counted by the oracle:
# tests/test_oracle_equirect.py) + the stores.  The honest yardstick for a kernel that shares reads between views (`roofline.frac_union`).
UNION_BYTES_PER_FRAME = 11_520_000 + 11_685_864 * 3
HBM_STREAM_GBS = 6290.0
# What the memory system delivers for this kernel's traffic SHAPE with no arithmetic at all (profiles/tools/membench.hip, profiles/r06/membench/:
# 784-byte row pieces at the frame's stride over 16 distinct 8K frames, one 16-byte store per five 16-byte loads): reads alone 6.1-6.3 TB/s,
# with the stores 4.8-5.5 TB/s over boxes, minutes, occupancy, depth and load instruction (registers or LDS copies); `frac_of_mix` is taken against
# the BEST of those.
MEMSYS_READ_ONLY_GBS = 6320.0
MEMSYS_MIX_GBS = (4800.0, 5520.0)
LAUNCHES_PER_STEP = 16             # a step = 16 launches of `--frames` frames at EVERY N (round-4 verdict: same step semantics at N = 1 and N > 1)
EQ_KERNEL_NAMES = {0: "eq_views_kernel<3>", 1: "eq_staged_kernel", 2: "eq_srcmajor_kernel"}


def memsys_probe(dev):
    """The memory-system probe of profiles/tools/membench.hip (built as lib/libgs360probe.so -- measurement aid, not product) run IN-PROCESS on
    the GPU the bench just used, right after the timed region: bare loads of the headline's 784-byte row pieces, and the same with one stored byte per
    five loaded (registers / LDS copies).  {mode: GB/s} or None (library absent, or a probe failed).  ~1 s."""
    import ctypes
    path = ROOT / "360cam-pgm-3dgs-tools_amd" / "lib" / "libgs360probe.so"
    if not path.exists():
        return None
    try:
        lib = ctypes.CDLL(str(path))
        fn = lib.gs360_membench
        fn.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        fn.restype = ctypes.c_int
        out = {}
        for mode in ("rows", "rowsmix", "dmamix"):
            o = (ctypes.c_double * 7)()
            rc = fn(mode.encode(), int(dev), 4, 2, 8, 9, 1, o)
            if rc != 0:
                print(f"bench.py: memory probe {mode} failed ({rc})", file=sys.stderr)
                return None
            out[mode] = round(o[0] * 1e3, 1)
        return out
    except Exception as e:      # the headline line must survive a failure here
        print(f"bench.py: memory probe failed: {e!r}", file=sys.stderr)
        return None


def device_state(bus_id):
    """clocks / power / partition modes of the GPU a rank sits on, read from the amdgpu driver's sysfs files (no child process: a process
    that has initialised the GPU must not exec): sclk and mclk as the driver reports them RIGHT NOW -- call it while the device is busy."""
    base = pathlib.Path("/sys/bus/pci/devices") / bus_id.lower()
    out = {}

    def rd(p):
        try:
            return p.read_text().strip()
        except OSError:
            return None
    for key, name in (("product", "product_name"), ("compute_partition", "current_compute_partition"), ("memory_partition", "current_memory_partition"),
                      ("perf_level", "power_dpm_force_performance_level")):
        v = rd(base / name)
        if v is not None:
            out[key] = v
    for key, name in (("fclk_mhz", "pp_dpm_fclk"), ("socclk_mhz", "pp_dpm_socclk")):      # the level marked '*': fabric / SoC clocks (HBM traffic crosses the fabric)
        v = rd(base / name)
        if v:
            for ln in v.splitlines():
                if ln.rstrip().endswith("*"):
                    try:
                        out[key] = float(ln.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
                    except (IndexError, ValueError):
                        pass
    try:
        hw = sorted((base / "hwmon").glob("hwmon*"))[0]
    except (OSError, IndexError):
        hw = None
    if hw is not None:
        for key, name, scale in (("sclk_mhz", "freq1_input", 1e-6), ("mclk_mhz", "freq2_input", 1e-6), ("power_w", "power1_input", 1e-6),
                                 ("power_cap_w", "power1_cap", 1e-6), ("temp_c", "temp2_input", 1e-3)):
            v = rd(hw / name)
            if v is not None:
                try:
                    out[key] = round(float(v) * scale, 1)
                except ValueError:
                    pass
    return out


def norm_yaw(a):
    a = ((a + 180.0) % 360.0) - 180.0
    return 180.0 if abs(a + 180.0) < 1e-6 else a


def view_table():
    return [(norm_yaw(i * 360.0 / N_VIEWS), 0.0, HFOV, HFOV, SIZE, SIZE) for i in range(N_VIEWS)]


_BASE = None


def synth_frame(np, k):
    """image B of SURVEY 8(d): smooth gradients + 64-px checker + per-pixel hash noise, rolled 13*k px."""
    global _BASE
    if _BASE is None:
        x = np.arange(W, dtype=np.uint32)[None, :]
        y = np.arange(H, dtype=np.uint32)[:, None]
        img = np.empty((H, W, 3), np.uint8)
        noise = (((x * np.uint32(2654435761)) ^ (y * np.uint32(40503))) >> np.uint32(27)).astype(np.uint8)
        img[..., 0] = ((x * 255) // W).astype(np.uint8) + noise
        img[..., 1] = ((y * 255) // H).astype(np.uint8) + noise
        img[..., 2] = ((((x >> 6) + (y >> 6)) & 1) * 96).astype(np.uint8) + noise
        _BASE = img
    return np.ascontiguousarray(np.roll(_BASE, 13 * k, axis=1))


def cpu_baseline(np, frame, budget_s=10.0):
    """Oracle ("port" of the same arithmetic, oracle/gs360_oracle.c) on this box's host cores.
    The thread count is picked by a short sweep (many-core hosts are not fastest with every core on a
    3.84-MPix job); `cores` reports the thread count actually used for the timed sample."""
    from oracle import orc
    orc.build()
    views = [orc.make_view(*v) for v in view_table()]
    ncpu = os.cpu_count() or 1
    cand = sorted({c for c in (ncpu, ncpu // 2, 64, 32, 16) if 1 <= c <= ncpu}, reverse=True)
    best, best_t = cand[0], None
    sweep = {}
    for c in cand:
        orc.equirect_views_u8(frame, views, threads=c)              # warm this team size
        t0 = time.perf_counter()
        for _ in range(3):
            orc.equirect_views_u8(frame, views, threads=c)
        dt = (time.perf_counter() - t0) / 3
        sweep[c] = dt
        if best_t is None or dt < best_t:
            best, best_t = c, dt
    n, t0 = 0, time.perf_counter()
    while True:
        outs = orc.equirect_views_u8(frame, views, threads=best)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 2000:
            break
    mpix = n * N_VIEWS * SIZE * SIZE / 1e6
    # SURVEY 8(d) asks for both ends: the same port on ONE thread (2 passes, ~1 s) next to the best team size
    t1 = time.perf_counter()
    for _ in range(2):
        orc.equirect_views_u8(frame, views, threads=1)
    dt1 = (time.perf_counter() - t1) / 2
    from gs360 import hostmem                     # (CPUs this process may use: affinity mask cut by a cgroup quota, if any)
    return {"value": round(mpix / dt, 2), "unit": "MPix/s", "cores": best, "kind": "port",
            "usable_cpus": hostmem.effective_cpus(),
            "sample": f"{n} passes of 1 frame x {N_VIEWS} views (same 8K->6x800^2 workload) in {dt:.1f} s; "
                      f"OpenMP over (view,row), {best} of {ncpu} host threads (best of {cand})",
            "single_thread": {"value": round(N_VIEWS * SIZE * SIZE / 1e6 / dt1, 2), "unit": "MPix/s", "cores": 1,
                              "sample": "2 passes of the same frame on one thread"},
            "all_threads": {"cores": ncpu, "value": round(N_VIEWS * SIZE * SIZE / 1e6 / sweep[ncpu], 2) if ncpu in sweep else None,
                            "unit": "MPix/s", "sample": "3 passes at every host thread"}}, outs


_JSON_FD = None     # set when stdout had to be parked on stderr (torch.distributed runs): the one JSON line goes here


def emit(line: dict) -> None:
    text = json.dumps(line) + "\n"
    if _JSON_FD is None:
        sys.stdout.write(text)
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, text.encode())


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _self_launch(n):
    """`python bench.py --gpus N` typed by hand (no WORLD_SIZE in the environment): start the N ranks with the same
    launcher the driver uses.  Runs before anything touches the GPU; the launcher is a CHILD process (never an exec)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(pathlib.Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def stream_mode(args, ctx, np, gs360, rank, world, barrier, info, dist, torch):
    """BASELINE.json configs[2] fed from the host: 600 frames dealt to the ranks, pinned -> H2D -> 12 views -> D2H."""
    from gs360.sharding import frames_for_rank
    from gs360.stream import FramePipeline
    hf = 104.2500326978036                       # fov_from_focal_mm(14, 36): the full360coverage preset (PC:614-626)
    layout = [(0, 0), (45, 30), (45, -30), (90, 0), (135, 30), (135, -30), (180, 0), (-135, 30), (-135, -30), (-90, 0), (-45, 30), (-45, -30)]
    size = args.stream_size
    specs = [(float(y), float(p), hf, hf, size, size) for y, p in layout]
    views = [gs360.View.make(*v) for v in specs]
    mine = frames_for_rank(args.stream_frames, world, rank)
    n_slots = 6
    # (four frames per launch: the ring family takes the source-major kernel, as in the resident job; the copies are per frame as before)
    pipe = FramePipeline(ctx, W, H, C, views, n_slots=n_slots, copy_out=False, batch=4)   # results alias pinned memory
    # every slot's pinned input holds its own synthetic frame; a decoder would write the next frame in place
    # (FramePipeline.acquire()/commit(), what gs360/video.py's reader does), so no host-side copy sits in the timed loop
    for k in range(n_slots):
        _done, buf = pipe.acquire()
        buf[:] = synth_frame(np, k + 7 * rank).reshape(-1)
        pipe.commit(tag=("warm", k))
    last = None
    for tag, outs in pipe.drain():
        last = (tag, outs)
    barrier()
    t0 = time.perf_counter()
    for fidx in mine:
        _done, _buf = pipe.acquire()
        pipe.commit(tag=fidx)
    for tag, outs in pipe.drain():
        last = (tag, [outs[k].copy() for k in (0, 1, 8)])
    ctx.sync(-1)
    local = time.perf_counter() - t0
    barrier()
    elapsed, _with_barrier = job_seconds(local, time.perf_counter() - t0, args, dist, torch)
    parity = None
    if rank == 0 and not args.no_cpu_baseline and last is not None:
        from oracle import orc
        orc.build()
        slot_frame = synth_frame(np, ((len(mine) - 1) % n_slots if mine else 0) + 7 * rank)   # what the last slot's buffer holds
        pick = [0, 1, 8]
        want = orc.equirect_views_u8(slot_frame, [orc.make_view(*specs[k]) for k in pick], threads=0)
        parity = all(np.array_equal(g, w) for g, w in zip(last[1], want))
    pipe.close()
    if rank == 0:
        px = args.stream_frames * len(views) * size * size
        in_b, out_b = W * H * C, len(views) * size * size * C
        line = {
            "metric": "MPix/s remapped, 8K equirect->preset views", "value": round(px / elapsed / 1e6, 1), "unit": "MPix/s",
            "n_gpus": world, "steps": args.stream_frames, "warmup": n_slots, "ms_per_step": round(elapsed * 1e3 / max(1, args.stream_frames), 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"HOST-FED stream (not the headline): {args.stream_frames} 7680x3840x3 frames dealt round-robin to {world} rank(s) -> "
                                   f"full360coverage 12x{size}x{size}, pinned host -> H2D per frame, one launch of 4 frames x 12 views (source-major kernel), D2H per frame "
                                   "(BASELINE.json configs[2]; PC:1049-1078)",
                       "frames_total": args.stream_frames, "frames_rank0": len(mine), "views": len(views), "device": info["name"],
                       "rank_devices": info["rank_devices"], "world_seen": info["world_seen"],
                       "parallelism": f"frames sharded x{world}, no collective", "parity_vs_oracle": parity,
                       "frames_per_s": round(args.stream_frames / elapsed, 1), "rank0_seconds": round(local, 4),
                       "pcie_bytes_per_frame": {"h2d": in_b, "d2h": out_b},
                       "pcie_bound_frames_per_s_per_gpu": round(63e9 / max(in_b, out_b), 1)},
            "roofline": None, "cpu_baseline": None,
        }
        emit(line)


def job_seconds(local, incl_barrier, args, dist, torch):
    """The job's time = the slowest rank's time from the common start (barrier + synchronize on every rank) to that rank's OWN
    completion (its last launch synchronised): MAX over ranks of `local`.  The closing barrier + synchronize follow immediately;
    the time including them (also MAX over ranks) is reported beside it -- on one rank the two are the same number, on N ranks
    they differ by the latency of one RCCL barrier, which is not part of the job (a 20-step job is 0.8 ms per rank at N = 8)."""
    if dist is None:
        return local, incl_barrier
    t = torch.tensor([local, incl_barrier], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0].item()), float(t[1].item())


FULL360 = [(0, 0), (45, 30), (45, -30), (90, 0), (135, 30), (135, -30), (180, 0), (-135, 30), (-135, -30), (-90, 0), (-45, 30), (-45, -30)]
HFOV_14MM = 104.2500326978036                    # fov_from_focal_mm(14, 36): the full360coverage preset (PC:614-626)


def job_mode(args, ctx, np, gs360, rank, world, barrier, info, dist, torch):
    """BASELINE.json configs[2], kernel-only: the job's frames are dealt to the ranks, each rank keeps its share resident in HBM
    (frame k = image B rolled 13 k px) and renders the 12 full360coverage views of every frame, 16 frames per launch."""
    from gs360.sharding import frames_for_rank
    size = args.job_size
    specs = [(float(y), float(p), HFOV_14MM, HFOV_14MM, size, size) for y, p in FULL360]
    views = [gs360.View.make(*v) for v in specs]
    mine = frames_for_rank(args.job_frames, world, rank)
    t_up = time.perf_counter()
    d_frames = [ctx.to_device(synth_frame(np, k)) for k in mine]
    t_up = time.perf_counter() - t_up
    batch = gs360.capi.MAX_FRAMES
    d_out = [ctx.alloc(size * size * C) for _ in range(batch * len(views))]
    calls = []
    for b0 in range(0, len(d_frames), batch):
        fr = d_frames[b0:b0 + batch]
        calls.append(ctx.make_equirect_call(fr, W, H, C, views, d_out[:len(fr) * len(views)], slot=0))
    first_call_ms = steady_call_ms = plan_rebuild_ms = plan_build_ms = 0.0
    if calls:                                             # the first call builds the geometry's plan (once per context), the second is a steady one
        t_first = time.perf_counter()
        calls[0]()
        ctx.sync(0)
        first_call_ms = (time.perf_counter() - t_first) * 1e3
        t_first = time.perf_counter()
        calls[0]()
        ctx.sync(0)
        steady_call_ms = (time.perf_counter() - t_first) * 1e3
        us0 = ctx.get_option("srcmajor_plan_build_us")    # a later geometry of the same context: the views 0.25 degrees wider, four frames
        plan_build_ms = us0 / 1e3
        wider = [gs360.View.make(v[0], v[1], v[2] + 0.25, v[3] + 0.25, v[4], v[5]) for v in specs]
        nfw = min(4, len(d_frames), batch)
        ctx.equirect_views_dev(d_frames[:nfw], W, H, C, wider, d_out[:nfw * len(views)], slot=0)
        ctx.sync(0)
        plan_rebuild_ms = (ctx.get_option("srcmajor_plan_build_us") - us0) / 1e3
    barrier()
    ctx.event_record(0, 0)
    t0 = time.perf_counter()
    for c in calls:
        c()
    ctx.event_record(0, 1)
    ctx.sync(-1)
    local = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ctx.event_elapsed_ms(0, 0, 1) if calls else 0.0
    per_rank = [round(local, 6)]
    elapsed, with_barrier = job_seconds(local, elapsed, args, dist, torch)
    if dist is not None:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        mine_t = torch.tensor([local], dtype=torch.float64, device=dev)
        all_t = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(all_t, mine_t)
        per_rank = [round(float(x.item()), 6) for x in all_t]
    parity = None
    if rank == 0 and not args.no_cpu_baseline and calls:
        from oracle import orc
        orc.build()
        last0 = (len(calls) - 1) * batch                  # the last batch is still in d_out: two views of its first frame
        want = orc.equirect_views_u8(synth_frame(np, mine[last0]), [orc.make_view(*specs[v]) for v in (1, 6)], threads=0)
        parity = all(np.array_equal(ctx.download(d_out[v], (size, size, C)), w) for v, w in zip((1, 6), want))
    if rank == 0:
        px = args.job_frames * len(views) * size * size
        emit({
            "metric": "MPix/s remapped, 8K equirect->preset views", "value": round(px / elapsed / 1e6, 1), "unit": "MPix/s",
            "n_gpus": world, "steps": args.job_frames, "warmup": 2, "ms_per_step": round(elapsed * 1e3 / max(1, args.job_frames), 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"RESIDENT JOB (not the headline): {args.job_frames} 7680x3840x3 frames resident in HBM, dealt round-robin to "
                                   f"{world} rank(s) -> full360coverage 12x{size}x{size}, {batch} frames per launch, kernel-only "
                                   "(BASELINE.json configs[2])",
                       "frames_total": args.job_frames, "frames_rank0": len(mine), "views": len(views), "device": info["name"],
                       "rank_devices": info["rank_devices"], "world_seen": info["world_seen"],
                       "resident_GB_rank0": round(len(mine) * W * H * C / 1e9, 1), "upload_s_rank0": round(t_up, 1),
                       "parallelism": f"frames sharded x{world}, no collective", "parity_vs_oracle": parity,
                       "frames_per_s": round(args.job_frames / elapsed, 1), "per_rank_seconds": per_rank,
                       "seconds_incl_closing_barrier": round(with_barrier, 6),
                       "rank0_kernel_us_per_frame": round(kernel_ms * 1e3 / max(1, len(mine)), 2),
                       # outside the timed region: the plan of the geometry is built by the first call of a context, every later call reuses it
                       "plan_build_ms": round(plan_build_ms, 2),      # the library's own clock around the build (+ the builder's scratch blocks and kernels: first build of the context)
                       "first_call_ms": round(first_call_ms, 2), "steady_call_ms": round(steady_call_ms, 2),
                       "plan_rebuild_ms": round(plan_rebuild_ms, 2),                                    # a second geometry's plan (scratch and kernels already there)
                       "rank0_eq_kernel": EQ_KERNEL_NAMES.get(ctx.get_option("last_eq_kernel"), "?")},
            "roofline": None, "cpu_baseline": None,
        })


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--frames", type=int, default=16, help="distinct HBM-resident frames per step (one launch; <= 16)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=INT",
                    help="context option for an A/B run (e.g. srcmajor_stage=1); results never depend on options")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed launches before the warm-up steps until the device has been busy this long (clock ramp); 0 = off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` array (the other BASELINE configs, N = 1 only)")
    ap.add_argument("--no-memsys", action="store_true", help="skip the in-process memory-system probe (roofline.memsys.measured_here, N = 1 only)")
    ap.add_argument("--with-torch", action="store_true", help="force the torch.distributed plumbing at N=1 too")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-over-ranks (gloo: control-flow tests on a box "
                         "with fewer GPUs than ranks; ranks then share devices round-robin)")
    ap.add_argument("--mode", default="resident", choices=["resident", "job", "stream"],
                    help="resident (default, the headline): a fixed job of steps x frames HBM-resident cfg2 frame renders dealt to the ranks; "
                         "job: BASELINE configs[2] kernel-only (600 resident 8K frames -> 12 x 1600^2, dealt to the ranks); "
                         "stream: host-fed cfg3 frames dealt to the ranks, PCIe-bound.  All three are strong scaling")
    ap.add_argument("--job-frames", type=int, default=600, help="--mode job: frames of the job (all ranks); 88.5 MB of HBM each")
    ap.add_argument("--job-size", type=int, default=1600, help="--mode job: view size (full360coverage: 1600)")
    ap.add_argument("--stream-frames", type=int, default=600, help="--mode stream: total frames of the job (all ranks)")
    ap.add_argument("--stream-size", type=int, default=1600, help="--mode stream: view size (full360coverage: 1600)")
    ap.add_argument("--stride-pad", type=int, default=0, help="experiments only: extra bytes per source row")
    ap.add_argument("--src-width", type=int, default=7680, help="experiments only: equirect width (height = width/2); "
                    "any value other than 7680 is NOT the BASELINE workload and is labelled as such")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.src_width != W:
        globals()["W"], globals()["H"] = args.src_width, args.src_width // 2

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        if args.backend == "nccl":                 # one rank per GPU over RCCL: refuse here instead of letting a rank die at the rendezvous
            import torch                           # (device_count() does not initialise the GPU)
            n_dev = torch.cuda.device_count()
            if args.gpus > n_dev:
                print(f"bench.py: rank {n_dev} needs GPU {n_dev} but only {n_dev} are visible (one rank per GPU over RCCL); "
                      "--backend gloo lets ranks share devices for control-flow tests", file=sys.stderr)
                sys.exit(3)
        sys.exit(_self_launch(args.gpus))          # nothing has touched the GPU yet; the ranks are child processes
    world = int(env_world or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; run\n"
              f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
              f"--master-port P bench.py --gpus {args.gpus} ...\n(or plain `python bench.py --gpus {args.gpus}`, which starts it)",
              file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.with_torch
    dist = torch = None
    if use_dist:
        # torch first: its bundled HIP runtime has the same SONAME as the system one, so the engine
        # library then binds to the runtime already in the process instead of loading a second copy.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        n_dev = max(1, torch.cuda.device_count())
        if args.backend == "gloo":
            local_rank %= n_dev
        elif local_rank >= n_dev:
            print(f"bench.py: rank {rank} needs GPU {local_rank} but only {n_dev} are visible (one rank per GPU over RCCL); "
                  "--backend gloo lets ranks share devices for control-flow tests", file=sys.stderr)
            sys.exit(3)
        torch.cuda.set_device(local_rank)
        # RCCL prints banner lines on stdout ("Librccl path : ...", version) whenever a communicator comes up -- at init,
        # lazily at the first collective, sometimes at teardown.  stdout must carry exactly ONE JSON line, so fd 1 points at
        # stderr for the rest of the process and the JSON line is written to the saved descriptor (emit()).
        sys.stdout.flush()
        global _JSON_FD
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist.barrier()
        torch.cuda.synchronize()

    import numpy as np
    import gs360

    ctx = gs360.Context(device=local_rank if use_dist else 0, n_slots=3 if args.mode == "stream" else 2)
    for kv in args.option:                                # kernel-selection switches for A/B runs (include/gs360.h: gs360_ctx_set_option)
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    info = ctx.info()
    # which GPU each rank really sits on: every rank's PCI bus id is gathered into config.rank_devices (+ the world size the process
    # group reports), and a job whose RCCL ranks share a device is refused (exit 3) -- N ranks over RCCL must mean N distinct GPUs.
    # (gloo ranks may share devices: the control-flow tests on a 1-GPU box; the duplicates are reported, not refused)
    info["rank_devices"] = [ctx.pci_bus_id()]
    info["world_seen"] = 1
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, info["rank_devices"][0])
        info["rank_devices"] = [str(g) for g in gathered]
        info["world_seen"] = int(dist.get_world_size())
        if args.backend == "nccl" and len(set(info["rank_devices"])) != world:
            if rank == 0:
                print(f"bench.py: {world} RCCL ranks on {len(set(info['rank_devices']))} distinct device(s) {info['rank_devices']}: "
                      "one rank per GPU is the contract", file=sys.stderr)
            sys.exit(3)

    def barrier():
        ctx.sync(-1)
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    if args.mode == "stream":
        stream_mode(args, ctx, np, gs360, rank, world, barrier, info, dist if use_dist else None, torch)
        ctx.close()
        if use_dist:
            dist.destroy_process_group()
        return
    if args.mode == "job":
        job_mode(args, ctx, np, gs360, rank, world, barrier, info, dist if use_dist else None, torch)
        ctx.close()
        if use_dist:
            dist.destroy_process_group()
        return
    from gs360.sharding import frames_for_rank
    views = [gs360.View.make(*v) for v in view_table()]
    nf = max(1, min(args.frames, gs360.capi.MAX_FRAMES))
    frames_host = [synth_frame(np, k + 7 * rank) for k in range(nf)]
    stride = W * C + args.stride_pad
    if args.stride_pad:
        padded = []
        for f in frames_host:
            buf = np.zeros((H, stride), np.uint8)
            buf[:, :W * C] = f.reshape(H, W * C)
            padded.append(buf)
        d_frames = [ctx.to_device(b) for b in padded]
    else:
        d_frames = [ctx.to_device(f) for f in frames_host]
    d_out = [ctx.alloc(SIZE * SIZE * C) for _ in range(nf * N_VIEWS)]
    step = ctx.make_equirect_call(d_frames, W, H, C, views, d_out, slot=0, src_stride=stride if args.stride_pad else 0)
    # the fixed job: `--steps` steps of LAUNCHES_PER_STEP launches of nf frames, the launches dealt round-robin to the ranks -- the same
    # step at every N (one launch is 0.25 ms: at the driver's --steps 20 the timed region is 78 ms on one rank, 10 ms per rank at
    # N = 8); `value` is a rate, comparable across N
    launches_per_step = LAUNCHES_PER_STEP
    n_launches = len(frames_for_rank(args.steps * launches_per_step, world, rank))
    n_mine = n_launches * nf
    # settle: the same launches, untimed, until the device has been busy for --settle-ms (clocks ramp over the first ~100 ms of
    # load: with 5 warm-up steps = 1.6 ms alone the 20 timed steps that follow run 6 % slower than in steady state), then the
    # caller's W warm-up steps
    t_plan = time.perf_counter()
    step()                                     # (first call apart: it builds the source-major plan; config.plan_build_ms = this call - a steady one)
    ctx.sync(0)
    first_call_ms = (time.perf_counter() - t_plan) * 1e3
    # ... and what a LATER geometry of this context costs (the first build also allocates the builder's scratch blocks and loads its
    # kernels): the same ring with a field of view 0.25 degrees wider, one frame, rendered once into the same outputs
    us0 = ctx.get_option("srcmajor_plan_build_us")
    plan_build_ms = us0 / 1e3
    wider = [gs360.View.make(v[0], v[1], v[2] + 0.25, v[3] + 0.25, v[4], v[5]) for v in view_table()]
    ctx.equirect_views_dev(d_frames[:1], W, H, C, wider, d_out[:N_VIEWS], slot=0, src_stride=stride if args.stride_pad else 0)
    ctx.sync(0)
    plan_rebuild_ms = (ctx.get_option("srcmajor_plan_build_us") - us0) / 1e3
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(8):
            step()
        ctx.sync(0)
    for _ in range(args.warmup):
        step()
    clocks = {"before": device_state(info["rank_devices"][rank if use_dist else 0])} if rank == 0 else None      # (the warm-up launches are still running)
    barrier()
    ctx.event_record(0, 0)
    t0 = time.perf_counter()
    for _ in range(n_launches):
        step()
    ctx.event_record(0, 1)
    if clocks is not None:
        clocks["during"] = device_state(info["rank_devices"][rank if use_dist else 0])      # (queued launches still running: sysfs reads, ~0.1 ms)
    ctx.sync(-1)
    local = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms_total = ctx.event_elapsed_ms(0, 0, 1)                  # HIP events on the launch stream
    eq_kernel = ctx.get_option("last_eq_kernel")                     # which equirect kernel the timed launches ran (0 gather, 1 LDS-staged, 2 source-major)
    kernel_ms = kernel_ms_total * nf / max(1, n_mine)                # per full launch of nf frames
    per_rank = [round(local, 6)]
    elapsed, with_barrier = job_seconds(local, elapsed, args, dist if use_dist else None, torch)
    if use_dist:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        mine_t = torch.tensor([local], dtype=torch.float64, device=dev)
        all_t = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(all_t, mine_t)
        per_rank = [round(float(x.item()), 6) for x in all_t]
    if n_launches == 0:
        step()                                                       # leave a full batch in d_out for the check below
        ctx.sync(-1)

    # correctness spot-check of what was timed (rank 0): frame 0 against the oracle, plus the CPU baseline
    cpu = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        if world == 1:
            cpu, want = cpu_baseline(np, frames_host[0])
        else:   # the timed CPU sample is an N=1 item; at N>1 rank 0 only checks what it rendered (one oracle pass)
            from oracle import orc
            orc.build()
            want = orc.equirect_views_u8(frames_host[0], [orc.make_view(*v) for v in view_table()], threads=0)
        got = [ctx.download(d_out[k], (SIZE, SIZE, C)) for k in range(N_VIEWS)]
        parity = all(np.array_equal(g, w) for g, w in zip(got, want))
        if not parity:
            print("bench.py: GPU output differs from the oracle -- number is INVALID", file=sys.stderr)
    # this box's own reading of the memory system, seconds after the timed region (N = 1 only; --no-memsys skips it: the profiled runs)
    memsys_here = None
    if rank == 0 and world == 1 and not args.no_memsys and args.mode == "resident":
        ctx.sync(-1)
        memsys_here = memsys_probe(0)
    # the other BASELINE configs on the same record (N = 1 only, after the headline's timed region and buffers are done with):
    # cfg1 / cfg3 / cfg5 (+ mask), the cubic default, cfg4 -- same timing method, one view of each against the oracle
    secondary = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_secondary and W == 7680 and args.stride_pad == 0:
        for b in d_frames + d_out:
            ctx.free(b)
        d_frames, d_out = [], []
        sys.path.insert(0, str(ROOT / "tests"))
        sys.path.insert(0, str(ROOT / "tests" / "tools"))
        try:
            import bench_configs
            t_sec = time.perf_counter()
            secondary = bench_configs.secondary_rows(ctx, steps=20)
            print(f"bench.py: secondary configs took {time.perf_counter() - t_sec:.1f} s", file=sys.stderr)
        except Exception as e:      # the headline line must survive a failure here
            print(f"bench.py: secondary configs failed: {e!r}", file=sys.stderr)
            secondary = [{"error": repr(e)}]

    if rank == 0:
        px_per_step = launches_per_step * nf * N_VIEWS * SIZE * SIZE
        ms_per_step = elapsed * 1e3 / max(1, args.steps)
        value = px_per_step * args.steps / elapsed / 1e6             # the whole fixed job / the slowest rank's time
        baseline_shape = (W == 7680 and args.stride_pad == 0)   # ALGO_BYTES_PER_FRAME was counted for the 7680-wide source only
        algo_bytes = ALGO_BYTES_PER_FRAME * nf
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        line_bytes = UNION_LINE_BYTES_PER_FRAME if eq_kernel == 2 else LINE_BYTES_PER_FRAME
        traffic = None
        tf = ROOT / "profiles" / "hbm_traffic.json"        # written from rocprofv3 --pmc passes (see profiles/README.md)
        if tf.exists() and baseline_shape:
            try:
                rec = json.loads(tf.read_text())
                if rec.get("frames_per_launch") == nf and rec.get("kernel", "").startswith(EQ_KERNEL_NAMES.get(eq_kernel, "?").split("<")[0]):
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "MPix/s remapped, 8K equirect->preset views",
            "value": round(value, 1), "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("" if W == 7680 else f"EXPERIMENT {W}x{H} source, NOT the baseline workload: ") +
                                   "7680x3840x3 u8 equirect -> --preset default --count 6 --size 800 (6x800x800), "
                                   "bilinear 1/32-px fixed point (BASELINE.json configs[1])",
                       "frames_per_step": nf, "views": N_VIEWS, "out_px_per_step": px_per_step,
                       "device": info["name"], "rank_devices": info["rank_devices"], "world_seen": info["world_seen"],
                       "parallelism": f"frames sharded x{world}, no collective",
                       "launches_per_step": launches_per_step,
                       "job": f"{args.steps} steps x {launches_per_step} launch(es) x {nf} frames = {args.steps * launches_per_step * nf} frame renders, "
                              f"launches dealt round-robin to {world} rank(s)",
                       "frames_rank0": n_mine, "per_rank_seconds": per_rank, "seconds_incl_closing_barrier": round(with_barrier, 6),
                       "settle_ms": args.settle_ms,
                       # the first call of the geometry (plan built once per context and geometry, outside the timed region) minus one steady launch
                       "plan_build_ms": round(plan_build_ms, 2),      # the library's own clock around the build (+ the builder's scratch blocks and kernels: first build of the context)
                       "first_call_ms": round(first_call_ms, 2),                                        # ... inside the first call (+ code-object load, first launch)
                       "plan_rebuild_ms": round(plan_rebuild_ms, 2),                                    # a second geometry's plan (scratch and kernels already there)
                       "clocks": clocks,
                       "parity_vs_oracle": parity},
            "roofline": None if not baseline_shape else {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         # same launch time against the bytes the PMC counters saw move (whole 128-B lines), for context
                         "traffic_frac": (round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
                         # ... and against what loads and stores of this shape and ratio reach without any arithmetic (membench, round 6)
                         "memsys": {"what": "784-byte row pieces of 16 distinct 8K frames + one stored byte per five loaded, no arithmetic (profiles/r06/membench/)",
                                    "read_only": MEMSYS_READ_ONLY_GBS, "mix_5_to_1": list(MEMSYS_MIX_GBS), "unit": "GB/s",
                                    "traffic_rate": (round(traffic / (kernel_ms * 1e-3) / 1e9, 1) if traffic else None),
                                    "frac_of_mix": (round(traffic / (kernel_ms * 1e-3) / 1e9 / MEMSYS_MIX_GBS[1], 4) if traffic else None),
                                    # ... and THIS box's reading, taken in-process right after the timed region (lib/libgs360probe.so = profiles/tools/membench.hip):
                                    # row pieces alone / + the 1 : 5 stores through registers / through LDS copies (the kernel's own load instruction)
                                    "measured_here": memsys_here,
                                    "frac_of_mix_here": (round(traffic / (kernel_ms * 1e-3) / 1e9 / max(memsys_here["rowsmix"], memsys_here["dmamix"]), 4)
                                                         if traffic and memsys_here else None)},
                         "kernel": EQ_KERNEL_NAMES.get(eq_kernel, str(eq_kernel)), "kernel_ms": round(kernel_ms, 5),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         # SURVEY 8(d)'s stricter figure (union of the views' texels, each once, + stores): what a kernel that shares reads across views is held to
                         "union_bytes_per_launch": UNION_BYTES_PER_FRAME * nf,
                         "frac_union": round(UNION_BYTES_PER_FRAME * nf / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         # the reachable bound next to the algorithmic one: distinct 128-B lines per view + stores at the
                         # measured streaming rate; frac = launch time at that bound / measured launch time
                         "line_bound": {"bytes_per_launch": line_bytes * nf, "peak": HBM_STREAM_GBS, "unit": "GB/s",
                                        "what": ("union of the six views' lines, each once" if eq_kernel == 2 else "distinct lines per view, summed") + " + stores",
                                        "achieved": round(line_bytes * nf / (kernel_ms * 1e-3) / 1e9, 1),
                                        "frac": round(line_bytes * nf / (kernel_ms * 1e-3) / 1e9 / HBM_STREAM_GBS, 4)}},
            "cpu_baseline": cpu,
        }
        if secondary is not None:
            line["secondary"] = secondary
        emit(line)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
