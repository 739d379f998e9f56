#!/usr/bin/env python3
"""Benchmark of the PBR frame path on MI355X.

Default mode (`--mode orbit`): a step is one 3840x2160 DamagedHelmet PBR+IBL frame (vertex transform, binning, tile
raster + skybox, surface shading) rendered into device memory with every input already resident in HBM.  With
--gpus N > 1 the frames of a camera orbit (one `updateCameraRotation(-10, 0)` step per frame) are dealt round-robin
to the ranks, one process per GPU, no data-path collective: weak scaling.  Each rank keeps --in-flight frames in
flight (default 4): one renderer context plus forks of it, which share ONE copy of the scene on the device and each
own a stream and a colour target, take the rank's frames in turn, so the short latency-bound geometry kernels of
one frame overlap the shading of the previous one.  `config.single_frame_ms` is the same frame with nothing else in
flight, on the context alone (its forks destroyed).

`--mode strips`: the synthetic stress scene (10 x 10 instanced helmets, 7680x4320).  Every rank renders its
even-aligned band of rows of every frame and the bands are gathered to rank 0 over RCCL: strong scaling of one frame.

`python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run on 127.0.0.1).

Prints ONE JSON line on rank 0.  `roofline` is for the surface-shading kernel: its launch duration measured with
HIP events on the stream it runs on, with the chip to itself, in the shape the library launches while the context
has forks, i.e. the one of the timed region (`roofline.in_flight` has the same kernel inside the timed region, where
the other frames' kernels stretch it; `roofline.lone_context` the shape a context without forks gets).  `cpu_baseline` is the CPU oracle (a restatement of the
reference WGSL, not Dawn/SwiftShader) rendering frames of the same orbit on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak


def _newest(*names):
    for n in names:
        p = os.path.join(ROOT, "profiles", n)
        if os.path.isfile(p):
            return p
    return os.path.join(ROOT, "profiles", names[-1])


TRAFFIC_FILE = _newest("r03_traffic.json", "r02_traffic.json", "r01_traffic.json")   # PMC pass of scripts/gpu_traffic.sh over this same workload
VALU_FILE = _newest("r03_valu.json", "r02_valu.json", "r01_valu.json")                # SQ_INSTS_VALU per launch, from scripts/gpu_pmc_frame.sh
PREFILTER_COUNTER_FILE = _newest("r03_prefilter_counters.txt", "r02_prefilter_counters.txt")   # scripts/gpu_pmc_prefilter.sh: counters per launch of the prefilter kernels
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X: 256 CUs x 4 SIMDs x 16 lanes x 2 (packed) x 2 (fma) x 2.4 GHz
PREFILTER_FLOP_PER_SAMPLE = 110   # the survey's fixed count for one sample of prefiltermap.wgsl:131-168 that passes N.L > 0
PREFILTER_ISSUE_CYCLES_PER_INSTRUCTION = 2.94  # the prefilter's two sample loops' compiled instruction mix priced with the measured issue rates (scripts/isa_issue_cost.py)
XCDS = 8
FRAME_KERNELS = ["setup_kernel", "raster_kernel", "resolve_surface_fast_kernel"]
# Vector instructions do not all cost the same issue time: plain f32 add / mul / fma and the simple integer operations
# retire at about one per 2.3 cycles per SIMD, anything that reads a scalar register, converts, compares or selects at
# one per 4, the transcendental unit's at one per 8 (scripts/valu_rate_probe.hip, profiles/r02_valu_issue_rates.jsonl);
# VALU_FILE holds each kernel's mean over its compiled instruction mix.  4.0 is the fallback for a file without it.
SIMDS, CUS, PEAK_CLOCK_HZ, CYCLES_PER_VALU_INSTRUCTION = 1024, 256, 2.4e9, 4.0


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed counter measurement (the benchmark
    itself runs without a profiler); None if there is none."""
    try:
        with open(TRAFFIC_FILE) as f:
            table = json.load(f)
        if kernel in table:
            return float(table[kernel]["hbm_bytes"])
        if "<" in kernel and not any(k.startswith(kernel.split("<")[0] + "<") for k in table):
            kernel = kernel.split("<")[0]           # a table that does not tell the instantiations apart
            if kernel in table:
                return float(table[kernel]["hbm_bytes"])
        # a templated kernel is listed once per instantiation ("name<256>", "name<1024>"): the one launched most often
        # in the measured run is the one the timed region launches
        shapes = [v for k, v in table.items() if k.startswith(kernel + "<")]
        return float(max(shapes, key=lambda v: v["launches"])["hbm_bytes"]) if shapes else None
    except (OSError, KeyError, ValueError):
        return None


def surface_shape(contexts_in_flight, views=1):
    """Name of the surface kernel's instantiation the library launches: small workgroups for the single frames of a
    context with forks, large ones for a context on its own and for every pass of several views."""
    return "resolve_surface_fast_kernel<%d>" % (256 if contexts_in_flight > 1 and views == 1 else 1024)


def pass_kernels(views, contexts_in_flight):
    """The three launches of a pass of `views` frames as the traffic table names them."""
    if views > 1:
        return ["setup_views_kernel", "raster_views_kernel", surface_shape(contexts_in_flight, views)]
    return ["setup_kernel", "raster_kernel", surface_shape(contexts_in_flight)]


def measured_frame_traffic(views=1, contexts_in_flight=2):
    """HBM-side bytes per frame: the pass's three launches, divided by the frames a pass draws."""
    parts = [measured_traffic(k) for k in pass_kernels(views, contexts_in_flight)]
    if any(p is None for p in parts):   # a table from before the kernels had these names
        parts = [measured_traffic(k) for k in FRAME_KERNELS]
        views = 1
    return None if any(p is None for p in parts) else float(sum(parts)) / views


def valu_issue(kernels, ms):
    """Shares of the chip's vector-instruction issue time, L1 tag lookups (one per compute unit per cycle) and
    texture-address-unit time that `kernels` use when they take `ms` per launch, from the counters of VALU_FILE and the
    live time, at the peak clock.  The frame kernels are priced against HBM bandwidth because the task says so; these
    are the units they actually keep busy, and none of them is saturated: the kernels are bound by the latency of
    their dependent load rounds at the few waves per SIMD their registers allow."""
    try:
        with open(VALU_FILE) as f:
            table = json.load(f)
        n = sum(float(table[k]["valu_wave_instructions"]) for k in kernels)
        cycles = sum(float(table[k]["valu_wave_instructions"]) * float(table[k].get("issue_cycles_per_instruction", CYCLES_PER_VALU_INSTRUCTION)) for k in kernels)
    except (OSError, KeyError, ValueError):
        return None
    clock_cycles = ms * 1e-3 * PEAK_CLOCK_HZ
    out = {"wave_instructions": n, "issue_cycles_per_instruction": cycles / n, "frac": cycles / (SIMDS * clock_cycles)}
    try:
        out["l1_tag_lookup_frac"] = sum(float(table[k]["l1_tag_lookups"]) for k in kernels) / (CUS * clock_cycles)
        out["texture_address_unit_frac"] = sum(float(table[k]["texture_address_busy_cycles"]) for k in kernels) / (CUS * clock_cycles)
    except KeyError:
        pass
    return out


def prefilter_counters(path=None):
    """The counters of the prefilter's dominant launch (the 1024-thread shape: levels 0 to 2 of the 2048^2 map in
    one launch, nine tenths of the time) from the committed rocprofv3 summary, per launch: vector instructions, the
    graphics-active cycles (summed over the eight XCDs by the profiler) and the texture-address units' busy cycles
    (summed over the 256 units).  Derived: the share of the vector issue slots the launch used at the clock it actually
    ran at, and the share of time the texture-address units were busy.  None if the file is missing or has another
    shape."""
    try:
        table, kernel = {}, None
        with open(path or PREFILTER_COUNTER_FILE) as f:
            for line in f:
                if " dispatches " in line:
                    kernel = "1024" if "1024>" in line or "prefilter_levels_kernel" in line else "other"   # (the large levels share one launch)
                    continue
                parts = line.split()
                if kernel and len(parts) >= 2 and parts[0][:1].isupper() and "_" in parts[0]:
                    table.setdefault(kernel, {}).setdefault(parts[0], float(parts[1]))
        k = table["1024"]
        cycles = k["GRBM_GUI_ACTIVE"] / XCDS                    # cycles the launch lasted, at the clock it ran at
        return {"valu_wave_instructions": k["SQ_INSTS_VALU"], "cycles": cycles, "ta_busy_frac": k["TA_TA_BUSY_sum"] / CUS / cycles,
                "valu_issue_frac": k["SQ_INSTS_VALU"] * PREFILTER_ISSUE_CYCLES_PER_INSTRUCTION / (SIMDS * cycles),
                "vector_memory_instructions": k.get("SQ_INSTS_VMEM_RD"), "l2_requests": k.get("TCP_TCC_READ_REQ_sum")}
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def prefilter_roofline(executed_samples, ms, counters):
    """The FP32-vector roofline entry of the prefilter: executed flop (samples that contribute x the survey's 110 flop
    per sample) over the time, against the vector peak; next to it what the counters say bounds the kernel."""
    tflops = executed_samples * PREFILTER_FLOP_PER_SAMPLE / (ms * 1e-3) / 1e12
    out = {"bound": "fp32_valu", "achieved": tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_VECTOR_PEAK_TFLOPS,
           "executed_flop": executed_samples * PREFILTER_FLOP_PER_SAMPLE, "flop_per_sample": PREFILTER_FLOP_PER_SAMPLE}
    if counters is not None:
        out["valu_issue"] = counters["valu_issue_frac"]
        out["ta_busy"] = counters["ta_busy_frac"]
    return out


def pass_of_step(i, count, views, contexts):
    """What step i of `count` steps submits when frames are drawn `views` to a pass and the passes go to `contexts`
    contexts in turn: (context, first frame, end frame) at every views-th step - the pass of that step's frames, the
    last one shorter if `count` is not a multiple - and None at the steps between.  Every frame of the `count` steps is
    drawn exactly once, in the region its step belongs to."""
    if i % views:
        return None
    return (i // views) % contexts, i, min(i + views, count)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="steps per rank in the timed region (default: 360 frames = one orbit; 20 in strips mode)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=("orbit", "strips"), default="orbit")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--skybox-size", type=int, default=1024)
    ap.add_argument("--in-flight", type=int, default=3,
                    help="frames in flight per GPU: one renderer context plus forks of it, each on its own stream, take the rank's frames in turn")
    ap.add_argument("--views", type=int, default=4,
                    help="frames per pass: each context draws this many consecutive frames of the orbit with one pbr_draw_views_device call (1..4)")
    ap.add_argument("--passes-only", action="store_true",
                    help="for counter runs: skip the single-frame measurements after the timed region, so that every launch of the frame kernels in the run is a whole pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefilter", action="store_true", help="skip the IBL prefilter timing (2048^2, 10 mips)")
    ap.add_argument("--prefilter-size", type=int, default=2048)
    ap.add_argument("--prefilter-mips", type=int, default=10)
    args = ap.parse_args(argv)
    strips = args.mode == "strips"
    if args.steps is None:
        args.steps = 20 if strips else 360
    if args.warmup is None:
        args.warmup = 3 if strips else 20
    if args.width is None:
        args.width = 7680 if strips else 3840
    if args.height is None:
        args.height = 4320 if strips else 2160
    return args


def launch_plan(args, env):
    """What this invocation has to do about ranks: ("run", world) when it is one rank of `world` (or the only one),
    ("spawn", n) when it was asked for n GPUs outside a launcher and has to start the ranks itself, ("error", text)
    when the launcher's world size contradicts --gpus."""
    world = env.get("WORLD_SIZE")
    if world is None:
        return ("spawn", args.gpus) if args.gpus > 1 else ("run", 1)
    world = int(world)
    if world != args.gpus:
        return ("error", f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    return ("run", world)


def spawn_ranks(n, argv):
    """Starts the n ranks with torch.distributed.run on this node and relays rank 0's JSON line."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stdout.write(proc.stdout)
        raise SystemExit(proc.returncode or 1)
    print(lines[-1], flush=True)
    raise SystemExit(0)


def cpu_baseline(pkg, renderer, W, H, budget_s=12.0):
    """Times the oracle on frames of the same orbit (raster + shade) on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    import parity_utils as pu
    lib = pkg._lib
    g = renderer.geometries[0]
    t = g.textures
    images = [pkg.load_image_bitmap(t[k]) for k in ("baseColorURI", "metallicRoughnessURI", "normalURI", "emissiveURI", "occlusionURI")]
    osc = pu.OracleScene(g, images, renderer.hdrs[0], renderer.skybox_size, renderer.irradiance_size, renderer.prefilter_size,
                         renderer.prefilter_mips, renderer.lut_size)
    # The IBL maps are inputs of the frame path; take the ones already computed instead of spending
    # minutes of CPU time on the precompute, which this baseline does not measure.
    osc._maps = {
        "sky": renderer.download_map(lib.MAP_SKYBOX), "irr": renderer.download_map(lib.MAP_IRRADIANCE),
        "pre": [renderer.download_map(lib.MAP_PREFILTER, m) for m in range(renderer.prefilter_mips)],
        "lut": renderer.download_map(lib.MAP_BRDF_LUT),
    }
    threads = ob.load().orc_num_threads()
    osc.draw(renderer.frame_uniforms(), W, H)       # untimed: first touch of the scene arrays
    frames, spent = 0, 0.0
    while spent < budget_s and frames < 2000:
        u = renderer.frame_uniforms()
        t0 = time.perf_counter()
        osc.draw(u, W, H)
        spent += time.perf_counter() - t0
        frames += 1
        renderer.camera.update_camera_rotation(-10.0, 0.0)
    return {"value": frames * W * H / 1e6 / spent, "unit": "Mpix/s", "cores": threads, "kind": "port",
            "sample": f"{frames} consecutive {W}x{H} frames of the same orbit (vertex + raster + shade) by the CPU oracle in {spent:.1f} s, "
                      f"OpenMP {threads} threads; IBL maps taken as inputs"}


def stress_instances(pkg, base_model):
    """The synthetic stress layout: a 10 x 10 grid of helmets 2.5 units apart in x and y."""
    mx = pkg.mathx
    return np.stack([mx.mul(mx.translate(mx.identity(), [(i - 4.5) * 2.5, (j - 4.5) * 2.5, 0.0]), base_model) for j in range(10) for i in range(10)])


def init_distributed(args, torch, dist):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PBR_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, barriers, reductions, the exchanges) with a
    # single rank, to rehearse it on a one-GPU box under torch.distributed.run
    use_dist = world > 1 or bool(os.environ.get("PBR_BENCH_FORCE_DIST"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The frame streams are created and used before RCCL starts, so that they hold hardware queues of their own (a
    # process has four by default): created after it, the three streams share queues with RCCL's and their kernels
    # serialise, 78 against 63 us per frame.
    frame_streams = [torch.cuda.Stream(dev) for _ in range(max(1, args.in_flight))]
    for st in frame_streams:
        with torch.cuda.stream(st):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize(dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    return world, rank, local_rank, use_dist, dev, frame_streams


def timed_region(torch, dist, use_dist, dev, steps, step_fn):
    """EXACTLY `steps` steps between a barrier + synchronize on both sides; returns the max over ranks, in seconds.
    A rank's clock stops when its own steps have completed (after its synchronize); the closing barrier follows, and the
    maximum over the ranks is what a clock stopped behind the barrier would read, less the barrier's own latency (a
    collective and a host wait, some 50-100 us, which is a tenth of a 20-step region)."""
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(i)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed


def check_lanes(lanes):
    """Frames drawn without statistics report a capacity overflow (set-up buffer, tile bins) only when the context is
    synchronised through the library: do that for every context, so that a wrong frame cannot pass as a number."""
    for rr, _, _ in lanes:
        rr.synchronize()


def run_orbit(args, pkg, torch, dist, ctxinfo):
    world, rank, local_rank, use_dist, dev, frame_streams = ctxinfo
    W, H = args.width, args.height
    K = max(1, args.in_flight)
    V = min(4, max(1, args.views))
    r = pkg.Renderer(width=W, height=H, device=local_rank, skybox_size=args.skybox_size)
    r.init()
    lanes = []                                # one context (the first owns the scene, the others are forks of it) + stream + colour targets (one per view of a pass) per pass in flight
    for k in range(K):
        rr = r if k == 0 else r.fork()
        rr.set_stream(frame_streams[k].cuda_stream)   # a real (non-null) stream, so torch events and syncs see the kernels
        lanes.append((rr, frame_streams[k], [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(V)]))
    lib = r._lib
    torch.cuda.set_stream(lanes[0][1])

    # Orbit: global frame f is rendered by rank f mod world; one -10 px tumble (1 degree) per frame.  The
    # uniform blocks (what drawFrame writes into its uniform buffers, renderer.ts:1828-1932) are host
    # arithmetic and are prepared before the timed region.
    n_local = args.warmup + args.steps
    for _ in range(rank):
        r.camera.update_camera_rotation(-10.0, 0.0)
    uniforms = []
    for _ in range(n_local):
        uniforms.append(r.frame_uniforms())
        for _ in range(world):
            r.camera.update_camera_rotation(-10.0, 0.0)

    def draw(rr, colors, lo, hi):
        """frames [lo, hi) of this rank on one context: one frame, or one pass of several views"""
        if hi - lo == 1:
            rr.draw_frame_device(colors[0].data_ptr(), uniforms=uniforms[lo])
        else:
            rr.draw_views_device(uniforms[lo:hi], [colors[v].data_ptr() for v in range(hi - lo)])

    def make_step(first, count):
        """step i of `count` steps that start at frame `first` (see pass_of_step)"""
        def step(i):
            submit = pass_of_step(i, count, V, K)
            if submit is not None:
                lane, lo, hi = submit
                draw(lanes[lane][0], lanes[lane][2], first + lo, first + hi)
        return step

    # Part of setting up, like Renderer.init(): every context draws one full pass once, so that the working buffers of
    # all its views exist (they are allocated at a context's first pass of that many views) whatever --warmup is; then
    # the W warm-up steps.
    if n_local >= V:
        for rr, _, colors in lanes:
            draw(rr, colors, 0, V)
        torch.cuda.synchronize(dev)
    warm = make_step(0, args.warmup)
    for i in range(args.warmup):
        warm(i)
    torch.cuda.synchronize(dev)
    # the dominant kernel inside the timed region: timed on the first context's stream, on the first quarter (at least
    # eight) of the passes that context renders: an event pair costs about 10 us of stream time
    n_passes = (args.steps + V - 1) // V
    pkg._lib.check(r._ctx, lib.pbr_profile_begin(r._ctx, max(8, (n_passes + K - 1) // K // 4)))
    elapsed = timed_region(torch, dist, use_dist, dev, args.steps, make_step(args.warmup, args.steps))
    frames, surf_total, span = C.c_int32(), C.c_float(), C.c_float()
    pkg._lib.check(r._ctx, lib.pbr_profile_end(r._ctx, C.byref(frames), C.byref(surf_total), C.byref(span)))
    check_lanes(lanes)

    # the dominant kernel with the chip to itself: same frames, first context only, one at a time.  The library launches
    # the surface kernel in one of two shapes, by whether the context shares its scene with forks (several frames in
    # flight) or not; while the forks exist this is the shape that ran in the timed region.
    def isolated(n, views):
        """n frames on the first context alone, `views` per pass; the profile counts launches of the surface kernel"""
        frames_, total_, span_ = C.c_int32(), C.c_float(), C.c_float()
        pkg._lib.check(r._ctx, lib.pbr_profile_begin(r._ctx, n))
        for lo in range(args.warmup, args.warmup + n, views):
            draw(r, lanes[0][2], lo, min(lo + views, args.warmup + n))
        pkg._lib.check(r._ctx, lib.pbr_profile_end(r._ctx, C.byref(frames_), C.byref(total_), C.byref(span_)))
        r.synchronize()
        return frames_, total_, span_

    n_iso = min(args.steps, max(V, min(args.steps, 20) // V * V))   # whole passes (a run of fewer steps than views has one shorter pass)
    iso_frames, iso_total, iso_span = isolated(n_iso, V)
    # all three launches of the same passes, each with the chip to itself (four events per pass)
    kern_frames, kern_total = C.c_int32(), (C.c_float * 3)()
    pkg._lib.check(r._ctx, lib.pbr_profile_begin_kernels(r._ctx, n_iso))
    for lo in range(args.warmup, args.warmup + n_iso, V):
        draw(r, lanes[0][2], lo, min(lo + V, args.warmup + n_iso))
    pkg._lib.check(r._ctx, lib.pbr_profile_end_kernels(r._ctx, C.byref(kern_frames), kern_total, None))
    r.synchronize()
    kern_ms = [kern_total[k] / max(kern_frames.value, 1) for k in range(3)]
    # the frame with nothing else in flight and no other context on the scene (config 2's "single frame"): the forks go first
    for rr, _, _ in lanes[1:]:
        rr.close()
    lanes = lanes[:1]
    if args.passes_only:
        lone_frames, lone_total, lone_span = C.c_int32(0), C.c_float(float("nan")), C.c_float(float("nan"))
    else:
        lone_frames, lone_total, lone_span = isolated(min(args.steps, 20), 1)

    # coverage for the bandwidth model: sampled on a second pass over a tenth of the timed views (a
    # statistics call synchronises, so it stays out of the timed region)
    stats = pkg._lib.FrameStats()
    cov_sum, tri_sum, n_cov = 0.0, 0, 0
    for i in ([] if args.passes_only else range(args.warmup, args.warmup + args.steps, max(1, args.steps // 10))):
        r.draw_frame_device(lanes[0][2][0].data_ptr(), uniforms=uniforms[i], stats=stats)
        cov_sum += stats.covered_pixels / float(W * H)
        tri_sum += stats.setup_triangles
        n_cov += 1
    c = cov_sum / max(n_cov, 1)

    result = None
    if rank == 0:
        npix = W * H
        nf = max(frames.value, 1)
        surf_ms = surf_total.value / nf
        frame_ms = elapsed / args.steps * 1e3       # wall time per frame of this rank, launch gaps and overlap included
        n_verts, n_idx = r.geometries[0].vertices.size // 8, r.geometries[0].indices.size
        # Algorithmic bytes.  Surface resolve, per covered pixel: 4 B visibility read + 4 B colour
        # write + one unique rgba8 texel from each of the 5 material textures.  Whole frame: vertex
        # and index streams once, visibility write + read and colour write per pixel, the 20 B of
        # texels per covered pixel and one rgba32f skybox texel per background pixel.
        surf_bytes_frame = c * npix * 28.0
        surf_bytes = V * surf_bytes_frame           # one launch of the surface kernel shades the V views of a pass
        frame_bytes = n_verts * 32 + n_idx * 2 + npix * 12.0 + c * npix * 20.0 + (1.0 - c) * npix * 16.0
        in_flight_gbs = surf_bytes / (surf_ms * 1e-3) / 1e9
        iso_ms = iso_total.value / max(iso_frames.value, 1)
        iso_gbs = surf_bytes / (iso_ms * 1e-3) / 1e9
        single_frame_ms = lone_span.value / max(lone_frames.value - 1, 1)
        lone_ms = lone_total.value / max(lone_frames.value, 1)
        frame_gbs = frame_bytes / (frame_ms * 1e-3) / 1e9
        frame_traffic = measured_frame_traffic(V, K)
        surf_traffic = measured_traffic(surface_shape(K, V))   # per launch, i.e. for the V views of a pass, in the shape of the timed region
        issue_one = valu_issue(["resolve_surface_fast_kernel"], iso_ms / V)   # the committed counts are per frame
        # the rasteriser's launch of a pass: it also shades the skybox tiles the set-up launch left (sky_share_setup of
        # them ride there); its algorithmic bytes are the colour it writes for those background pixels plus the quad
        # lists it leaves for the surface launch (8 B per uniform quad, 20 B per mixed one: at most 5 B per covered pixel)
        sky_share_raster = 1.0 - (0.10 if V > 1 else (0.55 if K == 1 else 0.35))
        raster_bytes = V * (sky_share_raster * (1.0 - c) * npix * 4.0 + c * npix * 5.0)
        raster_ms = kern_ms[1]
        raster_gbs = raster_bytes / (raster_ms * 1e-3) / 1e9 if raster_ms > 0 else float("nan")
        raster_name = "raster_views_kernel" if V > 1 else "raster_kernel"
        issue_frame = valu_issue(FRAME_KERNELS, frame_ms)
        result = {
            "metric": "shaded Mpix/s (DamagedHelmet 4K PBR+IBL)",
            "value": world * args.steps * npix / 1e6 / elapsed,
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"DamagedHelmet {W}x{H} single frame, {args.skybox_size}^2 skybox cube, irradiance 32^2, prefilter 128^2 x5, LUT 512^2, synthetic 4096x2048 HDR"
                                   + f"; step i renders frame i of a 1-degree-per-frame camera orbit that starts at the reference's default view, frame f on rank f mod {world}"
                                   + (f"; a rank's frames are submitted {V} consecutive ones per call (pbr_draw_views_device), {K} calls in flight" if V > 1 else ""),
                       "covered_fraction": c, "triangles": tri_sum // max(n_cov, 1), "parallelism": "replicas only" if world == 1 else f"frame-sharded x{world}",
                       "frames_in_flight_per_gpu": K * V, "passes_in_flight_per_gpu": K, "views_per_pass": V, "untimed_setup_passes_per_context": 1 if n_local >= V else 0, "single_frame_ms": single_frame_ms,
                       "single_frame_mpix_per_s": npix / 1e6 / (single_frame_ms * 1e-3),
                       "covered_mpix_per_s_of_surface_kernel": V * c * npix / 1e6 / (iso_ms * 1e-3)},
            "kernel_ms": {"resolve_surface_fast_kernel": iso_ms, "resolve_surface_fast_kernel_in_flight": surf_ms, "resolve_surface_fast_kernel_lone_context": lone_ms,
                          "one_context_frame_to_frame": span.value / max(nf - 1, 1)},
            # the launch duration is the kernel's alone (nothing else on the chip); inside the timed region the same
            # kernel shares the chip with the other frames' kernels and its event-to-event time is longer than a step
            "roofline": {"bound": "hbm", "kernel": "resolve_surface_fast_kernel", "launch_ms": iso_ms, "achieved": iso_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": iso_gbs / HBM_PEAK_GBS, "traffic": surf_traffic, "algorithmic_bytes": surf_bytes, "views_per_launch": V,
                         "bound_measured": "latency of dependent load rounds at four waves per SIMD; busiest unit: vector issue", "valu_issue": issue_one,
                         "lone_context": {"launch_ms": lone_ms, "achieved": surf_bytes_frame / (lone_ms * 1e-3) / 1e9, "frac": surf_bytes_frame / (lone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "note": "one frame per launch, in the kernel's other shape (one 1024-thread workgroup per compute unit), which the library launches for a context without forks"},
                         "in_flight": {"ms": surf_ms, "achieved": in_flight_gbs, "frac": in_flight_gbs / HBM_PEAK_GBS,
                                       "note": "event-to-event time of the kernel inside the timed region, stretched by the other frames in flight"}},
            "roofline_raster": {"bound": "hbm", "kernel": raster_name, "launch_ms": raster_ms, "achieved": raster_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": raster_gbs / HBM_PEAK_GBS, "traffic": measured_traffic(raster_name), "algorithmic_bytes": raster_bytes, "views_per_launch": V,
                                "note": "the pass's second launch, alone on the chip: tile rasteriser plus the skybox colour of the tiles the set-up launch did not take; "
                                        "as long as the surface launch, so the frame has two dominant kernels",
                                "bound_measured": "vector issue of the skybox arithmetic (62 instructions per pixel) and its 8 sixteen-byte loads per thread; the tiles with geometry wait on their per-triangle loops",
                                "valu_issue": valu_issue(["raster_kernel"], raster_ms / V)},
            "kernel_ms_alone_per_pass": {"setup": kern_ms[0], "raster": kern_ms[1], "surface": kern_ms[2]},
            "frame_roofline": {"bound": "hbm", "kernels": "all kernels of one frame (setup + sky, raster + sky, surface); wall time per frame with the configured frames in flight",
                               "achieved": frame_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frame_gbs / HBM_PEAK_GBS, "algorithmic_bytes": frame_bytes,
                               "traffic": frame_traffic,
                               "traffic_frac": None if frame_traffic is None else frame_traffic / (frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "note": "algorithmic_bytes is the nominal figure of the survey's model: it counts 8 B per pixel of visibility-buffer write + read that the "
                                       "timed path does not perform (the resolve works from quad lists) and one rgba32f texel per sky pixel that a magnified cube never "
                                       "fetches; traffic is what the L2s' memory-side counters measured per frame",
                               "bound_measured": "latency (set-up, rasteriser) and vector issue (skybox, surface); no unit saturated", "valu_issue": issue_frame},
        }
    return r, lanes, result


def run_strips(args, pkg, torch, dist, ctxinfo):
    """The stress scene across ranks: every rank uploads the whole instanced scene, draws rows strip_rows(H, rank, world)
    of every frame and sends them to rank 0 (RCCL gather); a step is one whole frame put together on rank 0."""
    world, rank, local_rank, use_dist, dev, frame_streams = ctxinfo
    import importlib
    parallel = importlib.import_module(pkg.__name__ + ".parallel")
    sharding = importlib.import_module(pkg.__name__ + ".sharding")
    W, H = args.width, args.height
    loader = pkg.glTFLoader(); loader.load(pkg.renderer.DEFAULT_MODEL)
    models = stress_instances(pkg, np.asarray(loader.get_geometries()[0].model, dtype=np.float32))
    r = pkg.Renderer(width=W, height=H, device=local_rank, skybox_size=args.skybox_size, instance_models=models)
    r.init()
    r.camera.update_camera_distance((5.0 - 30.0) * 100.0, 0.0)     # dolly out to 30 units: all hundred helmets in view
    stream = frame_streams[0]
    r.set_stream(stream.cuda_stream)
    torch.cuda.set_stream(stream)
    y0, y1 = sharding.strip_rows(H, rank, world)
    # Two strip targets (and on rank 0 two frames): while frame i's strips travel, frame i + 1 is rendered into the other
    # target.  The render runs on the context's stream, the gather behind an event on a second stream; a target is
    # reused only after the gather that read it (two frames earlier) has finished.  No host synchronisation per frame.
    strips = [torch.empty((y1 - y0, W, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
    frames_out = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(2)] if (use_dist and rank == 0) else [None, None]
    comm_stream = frame_streams[1] if len(frame_streams) > 1 else torch.cuda.Stream(dev)
    strip = strips[0]
    n_local = args.warmup + args.steps
    uniforms = []
    for _ in range(n_local):
        uniforms.append(r.frame_uniforms())
        r.camera.update_camera_rotation(-10.0, 0.0)
    rendered = [torch.cuda.Event() for _ in range(2)]
    gathered = [torch.cuda.Event() for _ in range(2)]
    timing = {"render": [], "gather": []}          # (begin, end) event pairs of the timed steps
    frame_box = [None]
    serial = bool(os.environ.get("PBR_STRIPS_SERIAL"))   # for comparison: render, wait, gather, wait (what round 2 timed)

    def step(i, timed=True):
        k = i & 1
        if i >= 2:
            stream.wait_event(gathered[k])                       # the gather of frame i - 2 has read this target
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if timed:
            ev[0].record(stream)
        r.draw_frame_device(strips[k].data_ptr(), y0=y0, y1=y1, uniforms=uniforms[i])
        if timed:
            ev[1].record(stream)
        rendered[k].record(stream)
        if serial:
            stream.synchronize()
        if use_dist:
            comm_stream.wait_event(rendered[k])
            with torch.cuda.stream(comm_stream):
                if timed:
                    ev[2].record(comm_stream)
                frame_box[0] = parallel.gather_strips(strips[k], H, W, rank, world, dst=0, out=frames_out[k])
                if timed:
                    ev[3].record(comm_stream)
                gathered[k].record(comm_stream)
        else:
            frame_box[0] = strips[k]
            gathered[k].record(stream)
        if serial:
            torch.cuda.synchronize(dev)
        if timed:
            timing["render"].append((ev[0], ev[1]))
            if use_dist:
                timing["gather"].append((ev[2], ev[3]))

    for i in range(args.warmup):
        step(i, timed=False)
    torch.cuda.synchronize(dev)
    elapsed = timed_region(torch, dist, use_dist, dev, args.steps, lambda i: step(args.warmup + i))
    r.synchronize()
    render_ms = sum(a.elapsed_time(b) for a, b in timing["render"]) / max(len(timing["render"]), 1)
    gather_ms = sum(a.elapsed_time(b) for a, b in timing["gather"]) / max(len(timing["gather"]), 1) if timing["gather"] else 0.0
    stats = pkg._lib.FrameStats()
    r.draw_frame_device(strip.data_ptr(), y0=y0, y1=y1, uniforms=uniforms[args.warmup], stats=stats)
    result = None
    if rank == 0:
        npix = W * H
        frame_ms = elapsed / args.steps * 1e3
        result = {
            "metric": "shaded Mpix/s (100x instanced DamagedHelmet 7680x4320, screen strips)",
            "value": args.steps * npix / 1e6 / elapsed, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": frame_ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic stress: 10x10 instanced DamagedHelmet ({models.shape[0] * r.geometries[0].indices.size // 3} triangles) at {W}x{H}, "
                                   f"{world} even-aligned row strips, one per rank, gathered to rank 0", "parallelism": f"screen strips x{world}",
                       "rank0_rows": [y0, y1], "rank0_setup_triangles": int(stats.setup_triangles), "rank0_covered_fraction_of_strip": stats.covered_pixels / float((y1 - y0) * W)},
            # device time of this rank's render and of the gather as rank 0 sees it (events on their streams); a step
            # that hid one behind the other takes about the longer of the two, not their sum
            "render_ms_rank0": render_ms, "gather_ms_rank0": gather_ms, "overlap_ms_rank0": max(0.0, render_ms + gather_ms - frame_ms),
            "pipelined": not serial,
            "exchange": "none (one rank)" if not use_dist else f"gather of {world} rgba8 bands ({npix * 4 / 1e6:.0f} MB per frame) to rank 0 over RCCL, received straight into the frame's rows; "
                                                                "frame i + 1 is rendered while frame i's bands travel (two targets, two streams)",
        }
    return r, [(r, stream, strip)], result


def run_prefilter(args, pkg, torch, dist, ctxinfo, r, result):
    """Second headline figure: the IBL GGX prefilter at 2048^2 x 10 mips x 1024 samples.  One GPU filters the whole map;
    with several ranks each filters its shard (equal cost, not equal texel counts) and one all-gather (RCCL) makes the
    map whole on every GPU."""
    world, rank, local_rank, use_dist, dev, frame_streams = ctxinfo
    lane_stream = frame_streams[0]           # the torch stream the first context runs on (run_orbit)
    import importlib
    parallel = importlib.import_module(pkg.__name__ + ".parallel")
    sharding = importlib.import_module(pkg.__name__ + ".sharding")
    lib = r._lib
    size, mips = args.prefilter_size, args.prefilter_mips
    level_texels = [6 * max(1, size >> m) ** 2 for m in range(mips)]
    texels = sum(level_texels)
    info = {"config": f"{size}^2 cube, {mips} mips, 1024 samples/texel", "n_gpus": world}
    try:
        how = os.environ.get("PBR_PREFILTER_EXCHANGE", "p2p")
        sent = [0]

        def once():
            if not use_dist:
                pkg._lib.check(r._ctx, lib.pbr_ibl_prefilter(r._ctx, size, mips, 1024, 0, 1))
            else:
                sent[0] = parallel.prefilter_sharded(r, size, mips, rank, world, dev, how=how, stream=lane_stream, comm_stream=frame_streams[1] if len(frame_streams) > 1 else None,
                                                         rehearse=True)
            torch.cuda.synchronize(dev)
        once()                                   # allocate + warm (the exchange's staging buffers too: they are kept)
        ms = timed_region(torch, dist, use_dist, dev, 1, lambda i: once()) * 1e3
        if use_dist:
            cuts = sharding.prefilter_cuts(size, mips, world)
            info["shard_texels"] = [b - a for a, b in zip(cuts, cuts[1:])]
            info["replicated_texels"] = cuts[0]
            info["exchange_bytes_per_rank"] = sharding.prefilter_exchange_bytes(size, mips, world, how)
            info["exchange_bytes_rank0_sent"] = sent[0]
            info["exchange"] = "none (one rank)" if world == 1 else (f"every rank filters the first {cuts[0]} texels (roughness 0) for itself; the other {(texels - cuts[0]) * 16 / 1e6:.0f} MB of the "
                                f"{texels * 16 / 1e6:.0f} MB map are cut by cost and exchanged ("
                                + ("one all-gather of equal chunks, staging kept between calls" if how != "p2p" else "peer to peer, in place")
                                + ", while the first texels are being filtered), included in ms")
        kept = sharding.prefilter_kept_samples(mips)
        executed = sum(k * t for k, t in zip(kept, level_texels))
        info.update({"ms": ms,
                     # work in the reference's units: every texel of every level loops over 1024 samples (prefiltermap.wgsl:131)
                     "reference_samples": texels * 1024, "reference_gsamples_per_s": texels * 1024 / (ms * 1e-3) / 1e9,
                     # work executed: roughness 0 collapses to one sample per texel and samples with N.L <= 0 contribute nothing
                     "executed_samples": executed, "executed_gsamples_per_s": executed / (ms * 1e-3) / 1e9, "kept_samples_per_level": kept,
                     "roofline": prefilter_roofline(executed, ms, prefilter_counters()),
                     "bound_measured": "vector issue: the counted launch used nine tenths of the issue slots of the clock it ran at (about 2.0 GHz under this load, "
                                       "against the 2.4 GHz peak the roofline is priced at), with the texture-address units three quarters busy beside it; "
                                       "see " + os.path.relpath(PREFILTER_COUNTER_FILE, ROOT)})
    except Exception as exc:                             # the frame benchmark above stands on its own
        info["error"] = f"{type(exc).__name__}: {exc}"
    if rank == 0 and result is not None:
        result["ibl_prefilter"] = info
    # restore the frame's prefilter map
    pkg._lib.check(r._ctx, lib.pbr_ibl_prefilter(r._ctx, r.prefilter_size, r.prefilter_mips, 1024, 0, 1))
    torch.cuda.synchronize(dev)


def main():
    args = parse_args()
    plan, what = launch_plan(args, os.environ)
    if plan == "error":
        raise SystemExit("bench.py: " + what)
    if plan == "spawn":
        spawn_ranks(what, sys.argv[1:])
    import torch
    import torch.distributed as dist
    ctxinfo = init_distributed(args, torch, dist)
    world, rank, local_rank, use_dist, dev, _ = ctxinfo

    import __graft_entry__ as entry
    pkg = entry.load_package()
    if args.mode == "strips":
        r, lanes, result = run_strips(args, pkg, torch, dist, ctxinfo)
    else:
        r, lanes, result = run_orbit(args, pkg, torch, dist, ctxinfo)
        if not args.no_prefilter:
            run_prefilter(args, pkg, torch, dist, ctxinfo, r, result)
        if rank == 0 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(pkg, r, args.width, args.height)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
    r.close()


if __name__ == "__main__":
    main()
