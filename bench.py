#!/usr/bin/env python3
"""Benchmark of the PBR frame path on MI355X.

A step is one 3840x2160 DamagedHelmet PBR+IBL frame (vertex transform, binning, tile raster + skybox,
surface shading) rendered into device memory with every input already resident in HBM.  With
--gpus N > 1 the frames of a camera orbit (one `updateCameraRotation(-10, 0)` step per frame) are
dealt round-robin to the ranks, one process per GPU, no data-path collective: weak scaling.  Each
rank keeps --in-flight frames in flight (default 3): that many renderer contexts, each on its own
stream with its own colour target, take the rank's frames in turn, so the short latency-bound
geometry kernels of one frame overlap the shading of the previous one.

Prints ONE JSON line on rank 0.  `roofline` is for the surface-shading kernel, timed with HIP events
on the stream it runs on over the timed region (where it shares the chip with the other frames in
flight); `roofline.isolated` is the same kernel on the same frames with nothing else running.
`cpu_baseline` is the CPU oracle (a restatement of the reference WGSL, not Dawn/SwiftShader)
rendering frames of the same orbit on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_traffic.json")   # PMC pass of scripts/gpu_traffic.sh over this same workload


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed counter measurement (the benchmark
    itself runs without a profiler); None if there is none."""
    try:
        with open(TRAFFIC_FILE) as f:
            return float(json.load(f)[kernel]["hbm_bytes"])
    except (OSError, KeyError, ValueError):
        return None


VALU_FILE = os.path.join(ROOT, "profiles", "r01_valu.json")   # SQ_INSTS_VALU per launch, from scripts/gpu_pmc_frame.sh
SIMDS, PEAK_CLOCK_HZ, CYCLES_PER_VALU_INSTRUCTION = 1024, 2.4e9, 4.0   # one vector instruction per SIMD every four cycles (measured)


def valu_issue(kernels, ms):
    """Share of the chip's vector-instruction issue slots that `kernels` fill when they take `ms` per launch:
    counted wave-instructions x 4 cycles / (1024 SIMDs x time x 2.4 GHz).  The frame kernels are priced against
    HBM bandwidth because the task says so; this is the limit they actually run into."""
    try:
        with open(VALU_FILE) as f:
            table = json.load(f)
        n = sum(float(table[k]["valu_wave_instructions"]) for k in kernels)
    except (OSError, KeyError, ValueError):
        return None
    return {"wave_instructions": n, "frac": n * CYCLES_PER_VALU_INSTRUCTION / (SIMDS * ms * 1e-3 * PEAK_CLOCK_HZ)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=360, help="frames per rank in the timed region (360 = one full orbit on one GPU)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--skybox-size", type=int, default=1024)
    ap.add_argument("--in-flight", type=int, default=3,
                    help="frames in flight per GPU: that many renderer contexts, each on its own stream, take the rank's frames in turn")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefilter", action="store_true", help="skip the IBL prefilter timing (2048^2, 10 mips)")
    ap.add_argument("--prefilter-size", type=int, default=2048)
    ap.add_argument("--prefilter-mips", type=int, default=10)
    return ap.parse_args()


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


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PBR_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, barriers, reductions, the prefilter's
    # all-gather) with a single rank, to rehearse it on a one-GPU box under torch.distributed.run
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

    import __graft_entry__ as entry
    pkg = entry.load_package()
    W, H = args.width, args.height
    K = max(1, args.in_flight)
    lib = None
    lanes = []                                # one renderer context + stream + colour target per frame in flight
    for _ in range(K):
        r = pkg.Renderer(width=W, height=H, device=local_rank, skybox_size=args.skybox_size)
        r.init()
        stream = frame_streams[len(lanes)]    # a real (non-null) stream, so torch events and syncs see the kernels
        r.set_stream(stream.cuda_stream)
        lanes.append((r, stream, torch.empty((H, W, 4), dtype=torch.uint8, device=dev)))
    r = lanes[0][0]
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

    def step(i):
        rr, _, color = lanes[i % K]
        rr.draw_frame_device(color.data_ptr(), uniforms=uniforms[i])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    # the dominant kernel is timed on the first context's stream, on the first quarter (at least eight) of
    # the frames that context renders in the timed region: an event pair costs about 10 us of stream time
    pkg._lib.check(r._ctx, lib.pbr_profile_begin(r._ctx, max(8, (args.steps + K - 1) // K // 4)))
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    frames, surf_total, span = C.c_int32(), C.c_float(), C.c_float()
    pkg._lib.check(r._ctx, lib.pbr_profile_end(r._ctx, C.byref(frames), C.byref(surf_total), C.byref(span)))
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # the dominant kernel once more with the chip to itself: same frames, first context only
    iso_frames, iso_total, iso_span = C.c_int32(), C.c_float(), C.c_float()
    n_iso = min(args.steps, 20)
    pkg._lib.check(r._ctx, lib.pbr_profile_begin(r._ctx, n_iso))
    for i in range(args.warmup, args.warmup + n_iso):
        r.draw_frame_device(lanes[0][2].data_ptr(), uniforms=uniforms[i])
    pkg._lib.check(r._ctx, lib.pbr_profile_end(r._ctx, C.byref(iso_frames), C.byref(iso_total), C.byref(iso_span)))

    # coverage for the bandwidth model: sampled on a second pass over a tenth of the timed views (a
    # statistics call synchronises, so it stays out of the timed region)
    stats = pkg._lib.FrameStats()
    cov_sum, tri_sum, n_cov = 0.0, 0, 0
    for i in range(args.warmup, args.warmup + args.steps, max(1, args.steps // 10)):
        r.draw_frame_device(lanes[0][2].data_ptr(), uniforms=uniforms[i], stats=stats)
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
        surf_bytes = c * npix * 28.0
        frame_bytes = n_verts * 32 + n_idx * 2 + npix * 12.0 + c * npix * 20.0 + (1.0 - c) * npix * 16.0
        surf_gbs = surf_bytes / (surf_ms * 1e-3) / 1e9
        iso_ms = iso_total.value / max(iso_frames.value, 1)
        iso_gbs = surf_bytes / (iso_ms * 1e-3) / 1e9
        frame_gbs = frame_bytes / (frame_ms * 1e-3) / 1e9
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
                                   + f"; step i renders frame i of a 1-degree-per-frame camera orbit that starts at the reference's default view, frame f on rank f mod {world}",
                       "covered_fraction": c, "triangles": tri_sum // max(n_cov, 1), "parallelism": "replicas only" if world == 1 else f"frame-sharded x{world}", "frames_in_flight_per_gpu": K},
            "kernel_ms": {"resolve_surface_fast_kernel": surf_ms, "one_context_frame_to_frame": span.value / max(nf - 1, 1)},
            "roofline": {"bound": "hbm", "kernel": "resolve_surface_fast_kernel", "achieved": surf_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": surf_gbs / HBM_PEAK_GBS, "traffic": measured_traffic("resolve_surface_fast_kernel"), "algorithmic_bytes": surf_bytes,
                         "isolated": {"ms": iso_ms, "achieved": iso_gbs, "frac": iso_gbs / HBM_PEAK_GBS,
                                      "frame_ms": iso_span.value / max(iso_frames.value - 1, 1),
                                      "valu_issue": valu_issue(["resolve_surface_fast_kernel"], iso_ms)}},
            "frame_roofline": {"bound": "hbm", "kernels": "all kernels of one frame (vertex, setup, raster+sky, surface); wall time per frame with the configured frames in flight", "achieved": frame_gbs,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frame_gbs / HBM_PEAK_GBS, "algorithmic_bytes": frame_bytes,
                               "valu_issue": valu_issue(["setup_kernel", "raster_kernel", "resolve_surface_fast_kernel"], frame_ms)},
        }

    if not args.no_prefilter:
        # second headline figure: the IBL GGX prefilter at 2048^2 x 10 mips x 1024 samples.  One GPU filters
        # the whole map; with several ranks each filters its texel range and one all-gather (RCCL) makes
        # the map whole on every GPU.
        texels = sum(6 * max(1, args.prefilter_size >> m) ** 2 for m in range(args.prefilter_mips))
        info = {"config": f"{args.prefilter_size}^2 cube, {args.prefilter_mips} mips, 1024 samples/texel", "n_gpus": world}
        try:
            import importlib
            parallel = importlib.import_module(pkg.__name__ + ".parallel")

            def run_prefilter():
                if not use_dist:
                    pkg._lib.check(r._ctx, lib.pbr_ibl_prefilter(r._ctx, args.prefilter_size, args.prefilter_mips, 1024, 0, 1))
                else:
                    parallel.prefilter_sharded(r, args.prefilter_size, args.prefilter_mips, rank, world, dev)
                torch.cuda.synchronize(dev)
            run_prefilter()                                   # allocate + warm
            if use_dist:
                dist.barrier()
            t0 = time.perf_counter()
            run_prefilter()
            if use_dist:
                dist.barrier()
            ms = (time.perf_counter() - t0) * 1e3
            if use_dist:
                tmax = torch.tensor([ms], dtype=torch.float64, device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                ms = float(tmax.item())
                info["exchange"] = f"all-gather of the {texels * 16 / 1e6:.0f} MB map included"
            info.update({"ms": ms, "algorithmic_tflops": texels * 1024 * 110 / (ms * 1e-3) / 1e12, "fp32_peak_tflops": 157.3 * world})
        except Exception as exc:                             # the frame benchmark above stands on its own
            info["error"] = f"{type(exc).__name__}: {exc}"
        if rank == 0:
            result["ibl_prefilter"] = info
        # restore the frame's prefilter map
        pkg._lib.check(r._ctx, lib.pbr_ibl_prefilter(r._ctx, r.prefilter_size, r.prefilter_mips, 1024, 0, 1))
        torch.cuda.synchronize(dev)

    if rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(pkg, r, W, H)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
    for rr, _, _ in lanes:
        rr.close()


if __name__ == "__main__":
    main()
