#!/usr/bin/env python3
"""Headline benchmark: path segments per second of the render loop on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one whole frame of BASELINE config 4, the configuration the metric is
quoted on: scenes/gpu_showcase.json at 1920x1080, 1024 spp, max depth 8 (the reference's
own scene file; there are no weights or datasets, and nothing synthetic).  The
frame is split over interleaved 32x32 tiles, one process per GPU; each rank renders
its tiles through the C ABI (pt_render_tiles_device), the per-tile framebuffers are
gathered on rank 0 over RCCL (torch.distributed "nccl" gather) and untiled there.  No
collective touches the data path before that gather.

value = millions of path segments (closest-hit queries = rays x bounces)
summed over all ranks and steps / wall time (max over ranks).  Primary samples/s is
reported beside it.  The 5 KB scene upload is inside the timed region; the output
stays in HBM.

roofline: the dominant kernel is ptk::trace_kernel.  It is VALU-issue bound, not HBM
bound (its path state lives in registers): "roofline" states its algorithmic HBM
bytes (58 B primary ray in + 24 B radiance out per sample) honestly against the
8 TB/s peak (a small fraction by design) and "roofline_fp64" states the binding
resource.  See DESIGN.md.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_NOFMA_TOPS = 39.3    # 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz, one op per lane (no FMA contraction)


def seg_flops(n_sphere: int, n_box: int, n_plane: int) -> float:
    """Algorithmic FP64 ops per segment, SURVEY.md 8(d) / BASELINE.md 5."""
    return 23.0 * n_sphere + 12.0 * n_box + 14.0 * n_plane + 150.0


# BASELINE.json's configs: scene, width, height, spp, max depth.  The default line is config 4, the one the metric is quoted on;
# the others are parity-test cases that --config turns into bench lines for the per-config roofline evidence (profiles/r03_c*).
CONFIGS = {
    "C1": ("example_simple", 256, 256, 16, 4),
    "C2": ("test_scene", 800, 600, 256, 8),
    "C3": ("metal_glass_room", 1920, 1080, 1024, 12),
    "C4": ("gpu_showcase", 1920, 1080, 1024, 8),
    "C5": ("test_comprehensive", 3840, 2160, 4096, 16),
}


def host_cpu_quota() -> int:
    """CPUs this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 128))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="one of BASELINE.json's configs instead of the default line (config 4); sets scene, size, spp and depth")
    ap.add_argument("--scene", default="gpu_showcase")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--spp-chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the bounded CPU-baseline sample")
    pre, _ = ap.parse_known_args()
    if pre.config:  # the config's scene, size, spp and depth become the defaults: an explicit --spp (the short PMC passes) still wins
        sc_, w_, h_, spp_, d_ = CONFIGS[pre.config]
        ap.set_defaults(scene=sc_, width=w_, height=h_, spp=spp_, depth=d_)
    args = ap.parse_args()
    config_id = next((k for k, v in CONFIGS.items() if (v[0], v[1], v[2], v[4]) == (args.scene, args.width, args.height, args.depth)), None)
    config_full = config_id is not None and CONFIGS[config_id][3] == args.spp  # (the PMC passes run a config at a fraction of its spp)

    import torch  # first: libptcore must resolve libamdhip64.so.7 to the copy torch already loaded
    import torch.distributed as dist

    from path_trace_golang_amd import capi, distributed, hip, scene

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print("bench.py --gpus %d was launched with WORLD_SIZE=%d: start it with torch.distributed.run "
              "--nproc-per-node %d (or plainly for --gpus 1)" % (args.gpus, world, args.gpus), file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: torch.cuda.is_available() is False", file=sys.stderr)
        return 2
    # PT_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks: every rank
    # renders on GPU (local_rank mod visible GPUs) and the tile gather goes through host memory.  The
    # default, and what the driver runs, is "nccl" = RCCL over xGMI with one GPU per rank.
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, ngpu)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # Steady-state rendering (frame after frame on one context, what this bench times): job buffers of 160 GiB, so that a
    # 1080p x 1024 spp frame takes 4 passes instead of 13 (each pass ends in the ragged tail of five persistent kernels,
    # ~1.7 ms).  The library's own default stays 48 GiB because a one-shot render pays ~6 ms of hipMalloc per GiB.
    os.environ.setdefault("PTCORE_L_BUDGET_MB", "163840")
    L = capi.load()
    ctx = capi.Context(devices=[dev_index])
    sc = scene.load(os.path.join(ROOT, "scenes", args.scene + ".json"))
    flat = hip.FlatScene(sc)
    cfg = hip.pt_config(hip.RenderConfig(args.width, args.height, args.spp, args.depth, args.seed, args.spp_chunk, 0))
    W, H = args.width, args.height

    shard = capi.PtShard(rank, world)
    ntl, ntx, nty = C.c_int32(), C.c_int32(), C.c_int32()
    capi.check(L.pt_shard_tiles(W, H, C.byref(shard), C.byref(ntl), C.byref(ntx), C.byref(nty)))
    s0 = capi.PtShard(0, world)
    ntl_max = C.c_int32()
    capi.check(L.pt_shard_tiles(W, H, C.byref(s0), C.byref(ntl_max), None, None))
    stride_tiles = ntl_max.value

    tiles = torch.zeros(stride_tiles * 4096, dtype=torch.uint8, device=dev)
    scratch = None
    frame = None
    packed = None
    if rank == 0:
        # the gather lands straight in one contiguous buffer (shard k at k * stride_tiles tiles): no copy before the untile
        packed = torch.empty(world * stride_tiles * 4096, dtype=torch.uint8, device=dev) if world > 1 else None
        scratch = list(packed.split(stride_tiles * 4096)) if world > 1 else None
        frame = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def untile(bufs, stride):
        src = bufs[0]
        if len(bufs) > 1:
            if bufs[0].data_ptr() != packed.data_ptr():  # rehearsal path: the buffers came through host memory
                torch.cat(bufs, out=packed)
            src = packed
        capi.check(L.pt_untile_device(ctx.handle, W, H, world, stride, C.c_void_p(src.data_ptr()), None,
                                      C.c_void_p(frame.data_ptr()), W * 4, None, C.c_void_p(stream.cuda_stream)))
        return frame

    gather_ms = [0.0]  # rank 0, world > 1: device time of the tile gather + untile of the timed steps (torch events on the current stream)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if world > 1 else None
    pending = []

    def step():
        st = capi.PtStats()
        capi.check(L.pt_render_tiles_device(ctx.handle, C.byref(flat.c), C.byref(cfg), C.byref(shard),
                                            C.c_void_p(tiles.data_ptr()), None, C.c_void_p(stream.cuda_stream),
                                            C.byref(st)))
        if backend == "nccl" or world == 1:
            if ev is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
            distributed.assemble_frame(tiles, W, H, rank, world, untile, scratch)
            if ev is not None:
                b.record(stream)
                pending.append((a, b))
        else:  # rehearsal: gather through host memory
            host = tiles.cpu()
            bufs = distributed.gather_tiles(host, rank, world)
            if rank == 0:
                untile([b.to(dev) for b in bufs], stride_tiles)
        return st

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Set-up, outside the timed region whatever --warmup says: open and close a frame, which reserves the job buffers
    # (hipMalloc of the budget above costs about a second; a context keeps its buffers from frame to frame).
    capi.check(L.pt_begin(ctx.handle, C.byref(flat.c), C.byref(cfg)))
    capi.check(L.pt_end(ctx.handle, None))

    for _ in range(args.warmup):
        step()
    fence()
    pending.clear()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    gather_ms[0] = sum(a.elapsed_time(b) for a, b in pending) / max(1, args.steps)

    red_dev = dev if backend == "nccl" else torch.device("cpu")
    tot = torch.tensor([float(sum(s.segments for s in stats)), float(sum(s.samples for s in stats)),
                        float(sum(s.exit_scans for s in stats))], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    # per-rank device-busy time of the render (imbalance) next to the whole step (render + tile gather + untile)
    busy = torch.tensor([sum(s.device_ms for s in stats) / max(1, args.steps)], dtype=torch.float64, device=red_dev)
    busy_all = [busy]
    if world > 1:
        busy_all = [torch.zeros_like(busy) for _ in range(world)]
        dist.all_gather(busy_all, busy)
    busy_ms = [float(b.item()) for b in busy_all]
    segments, samples, exits = (float(x) for x in tot.tolist())

    out = None
    if rank == 0:
        steps = max(1, args.steps)
        trace_ms = sum(s.trace_ms for s in stats)
        glass_ms = sum(s.glass_ms for s in stats)
        resolve_ms = sum(s.resolve_ms for s in stats)
        chunk = stats[0].spp_chunk if stats else 0
        n_samples = float(sum(s.samples for s in stats))
        split_launches = sum(s.trace_split_launches for s in stats)
        if split_launches > 0:
            # dominant kernel: the split form of the trace kernel.  Its algorithmic HBM bytes: 58 B per fresh sample
            # (primary ray written by raygen_kernel: 6 doubles + 8 B stream state + 2 B draw count), 88 B per path taken
            # up from the continuation queue (9 doubles + stream state + job + depth), 24 B of radiance per path that
            # ends in it, 100 B per path it parks in the glass queue (10 doubles + stream state + job, depth, object)
            kernel_name = "ptk::trace_kernel<false,false,1,1>"  # <STATS, PROF, SCAN_BROAD, FORM_SPLIT>
            n_launch = split_launches
            k_ms = sum(s.trace_split_ms for s in stats)
            alg_total = (58.0 * n_samples + 88.0 * sum(s.split_cont_in for s in stats) + 24.0 * sum(s.split_finished for s in stats)
                         + 100.0 * sum(s.glass_events for s in stats))
            alg_note = ("register-resident paths between dielectric bounces: per launch 58 B per fresh sample in, 88 B per continuation in, "
                        "24 B radiance out per path ending, 100 B out per path parked for glass_kernel; the binding resource is VALU "
                        "issue, see roofline_fp64")
        else:
            kernel_name = "ptk::trace_kernel<false,false,1,0>"
            n_launch = sum(s.trace_launches for s in stats)
            k_ms = trace_ms
            alg_total = (58.0 + 24.0) * n_samples
            alg_note = ("register-resident paths: HBM carries 82 B per SAMPLE (primary ray in, radiance out), nothing per bounce; "
                        "the binding resource is VALU issue, see roofline_fp64")
        avg_launch_s = (k_ms / max(1, n_launch)) * 1e-3
        alg_bytes = alg_total / max(1, n_launch)
        achieved_gbs = alg_total / max(k_ms * 1e-3, 1e-12) / 1e9
        traffic = None
        traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        default_cfg = world == 1 and config_full and args.spp_chunk == 0
        if os.path.exists(tpath) and default_cfg:  # the PMC passes were taken on exactly this workload and launch shape
            try:
                with open(tpath) as f:  # (config 4 under the bare kernel name, the others under name@Cn)
                    tk = json.load(f).get(kernel_name if config_id == "C4" else kernel_name + "@" + config_id, {})
                # PMC bytes per algorithmic byte of the profiled launches x this run's algorithmic bytes per launch
                if tk.get("hbm_bytes_per_alg_byte"):
                    traffic = tk["hbm_bytes_per_alg_byte"] * alg_bytes
                    traffic_source = ("NOT measured in this run: %.3f HBM bytes per algorithmic byte from profiles/pmc_traffic.json [%s] x this "
                                      "run's algorithmic bytes per launch" % (tk["hbm_bytes_per_alg_byte"], tk.get("source", "?")))
            except Exception:
                traffic = None
        objs = [o.type for o in sc.objects]
        n_sph = sum(1 for t in objs if t in ("sphere", "sphere_light"))
        n_box = sum(1 for t in objs if t == "box")
        n_pl = sum(1 for t in objs if t == "plane")
        fseg = seg_flops(n_sph, n_box, n_pl)
        # kernel rate of the path work: every trace pass plus glass_kernel
        rank0_seg_per_s = sum(s.segments for s in stats) / max((trace_ms + glass_ms) * 1e-3, 1e-12)
        fp64_tops = rank0_seg_per_s * fseg / 1e12
        clocks = [s.shader_clock_mhz for s in stats if s.shader_clock_mhz > 0]
        clock_mhz = sum(clocks) / len(clocks) if clocks else None
        out = {
            "metric": "Msamples/s (rays x bounces/s) at %dx%dx%dspp" % (W, H, args.spp),
            "value": segments / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "scene file (scenes/%s.json, byte-identical to the reference's; nothing here is synthetic)" % args.scene,
            "config": {"workload": "scenes/%s.json %dx%d, %d spp, max depth %d, seed %d (%s)"
                                   % (args.scene, W, H, args.spp, args.depth, args.seed,
                                      ("BASELINE config %s" % config_id[1] + ("" if config_full else " at %d of its %d spp" % (args.spp, CONFIGS[config_id][3])))
                                      if config_id else "not a BASELINE config"),
                       "tiles": "32x32 interleaved over %d rank(s)" % world, "gather": ("rccl" if backend == "nccl" else backend + " (rehearsal)") if world > 1 else "none",
                       "spp_chunk": chunk, "job_buffer_budget_mib": int(os.environ["PTCORE_L_BUDGET_MB"])},
            "per_rank_render_ms_per_step": busy_ms,
            "per_rank_render_ms_min": min(busy_ms), "per_rank_render_ms_max": max(busy_ms),
            # N > 1: the tile gather (RCCL over xGMI, ~1 MB per peer at 1080p) + untile on rank 0, device time between two events on
            # the stream they run on (it includes the wait for the slowest rank's tiles); and what is left of the step
            "gather_untile_device_ms_per_step": gather_ms[0] if world > 1 else 0.0,
            "gather_untile_host_ms_per_step": elapsed / steps * 1e3 - max(busy_ms),
            "primary_msamples_per_s": samples / elapsed / 1e6,
            "segments_per_sample": segments / max(samples, 1.0),
            "exit_scans_per_segment": exits / max(segments, 1.0),
            "pixel_rmse_vs_cpu_ref": None,
            "timed_region": "K whole frames on a context whose job buffers exist (reserved at set-up): 5 KB scene upload, ray generation, trace, resolve, tile gather (N > 1) and untile; "
                            "the RGBA8 frame stays in HBM on rank 0, its 8.3 MB D2H copy (~0.2 ms) is excluded",
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved_gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "alg_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": avg_launch_s * 1e3, "launches_per_step": n_launch / steps,
                         "note": alg_note},
            "roofline_fp64": {"bound": "fp64_valu", "achieved": fp64_tops, "peak": FP64_PEAK_NOFMA_TOPS,
                              "unit": "Tflop/s (unfused)", "frac": fp64_tops / FP64_PEAK_NOFMA_TOPS,
                              "frac_of_fma_peak": fp64_tops / (2.0 * FP64_PEAK_NOFMA_TOPS),
                              # the clock the trace kernels actually held (pt_stats.shader_clock_mhz: shader cycles over the 100 MHz
                              # reference counter, one wave per launch) and the fraction of the unfused peak AT that clock
                              "shader_clock_mhz": clock_mhz,
                              "frac_at_measured_clock": (fp64_tops / (256 * 4 * 16 * clock_mhz * 1e6 / 1e12)) if clock_mhz else None,
                              "alg_flops_per_segment": fseg,
                              "trace_share_of_step": (trace_ms + glass_ms) / max(elapsed * 1e3, 1e-9),
                              "trace_ms_per_step": trace_ms / steps, "glass_ms_per_step": glass_ms / steps,
                              "glass_events_per_segment": sum(s.glass_events for s in stats) / max(1.0, float(sum(s.segments for s in stats))),
                              "raygen_ms_per_step": sum(s.raygen_ms for s in stats) / steps,
                              "resolve_ms_per_step": resolve_ms / steps},
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import ora  # CPU restatement of the reference engine: the timed baseline only

            import numpy as np

            workers = host_cpu_quota()
            osc = ora.Scene.load(os.path.join(ROOT, "scenes", args.scene + ".json"))
            # bounded sample of the same workload: full frame, 1 spp to size it, then ~cpu_seconds of CPU work
            r1 = ora.render(osc, W, H, 1, args.depth, seed=args.seed, workers=workers, want=("accum",))
            cpu_spp = int(max(1, min(64, round(args.cpu_seconds / max(r1["stats"]["seconds"], 1e-3)))))
            r = ora.render(osc, W, H, cpu_spp, args.depth, seed=args.seed, workers=workers, want=("rgba", "accum"))
            cst = r["stats"]
            out["cpu_baseline"] = {
                "value": cst["segments"] / cst["seconds"] / 1e6, "unit": "Msamples/s", "cores": cst["workers"],
                "kind": "port",
                "sample": "same scene and frame size, %d of %d spp (%.1f s of CPU work on %d threads = the box's CPU quota); "
                          "C restatement of internal/engine (oracle/pt_oracle.c): the Go reference cannot be built here"
                          % (cpu_spp, args.spp, cst["seconds"], cst["workers"]),
                "primary_msamples_per_s": cst["samples"] / cst["seconds"] / 1e6,
            }
            out["gpu_over_cpu"] = out["value"] / max(out["cpu_baseline"]["value"], 1e-12)
            # parity on the same sample: GPU at cpu_spp vs the CPU image
            img = np.zeros((H, W, 4), np.uint8)
            acc = np.zeros((H, W, 3), np.float64)
            hip.render(sc, hip.RenderConfig(W, H, cpu_spp, args.depth, args.seed), img, None, acc, ctx=ctx)
            g = np.sqrt(np.clip(acc / cpu_spp, 0, 1))
            o = np.sqrt(np.clip(r["accum"] / cpu_spp, 0, 1))
            out["pixel_rmse_vs_cpu_ref"] = float(np.sqrt(np.mean((g - o) ** 2)))
            out["rgba8_bytes_differing"] = int(np.count_nonzero(img != r["rgba"]))
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
