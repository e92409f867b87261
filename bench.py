#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X.

Metric (BASELINE.json): U-Net forward ms + active-block conv TFLOP/s vs
edit-ratio on the 256x256 DDPM (LSUN-Church) U-Net, fp32, synthetic activations,
random-init weights.  Workload at N=1: BASELINE.json configs[1] -- one sparse
(SIGE) forward of the DDPM-256 U-Net at a 1.2 % square edit.

    python bench.py --gpus N --steps K --warmup W

A "step" = one sparse U-Net forward per GPU (each rank edits its own region of
the shared original image; one process per GPU, RCCL).  The activation cache of
the original image is computed by rank 0 and RCCL-broadcast to the other ranks
in one flat buffer BEFORE the timed region (once per original image; its time is
reported as `cache_broadcast_ms`).  The timed region replays a hipGraph of the
sparse forward K times, bracketed by barrier + synchronize, max over ranks.

One JSON line on rank 0, with `roofline` (dominant hot-path kernel, measured with
HIP events on the launch stream, rotating buffers) and `cpu_baseline` (the
reference's own sige/cpu backend -- oracle/_ref -- under the same U-Net on the
host cores; falls back to the C restatement, kind "port", if _ref is absent).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_HBM_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak


def square_mask(ratio, H=256, W=256, top=100, left=90):
    side = int(round((ratio ** 0.5) * H))
    m = torch.zeros(H, W, dtype=torch.bool)
    m[top:top + side, left:left + side] = True
    return m


# ------------------------------------------------------------------ op trace --
TRACED = ("gather", "scatter_gather", "scatter_fused", "scatter_with_block_residual_fused", "block_conv",
          "block_conv_direct", "gather_conv", "scatter_gather_conv", "gather_conv_nchw",
          # channels-last forms (the layout the benchmark runs in)
          "gather_cl", "scatter_gather_cl", "scatter_cl", "scatter_with_block_residual_cl", "block_conv_cl",
          "gather_conv_cl", "scatter_gather_conv_cl", "scatter_gather_conv_scatter_cl")


class Tracer:
    """Wraps the sige_amd.hip entry points the modules call; while `log` is a list
    every call is recorded as (name, args, kwargs) holding the live tensors."""

    def __init__(self, hip):
        self.log = None
        for name in TRACED:
            orig = getattr(hip, name)

            def wrapped(*a, _orig=orig, _name=name, **k):
                if self.log is not None:
                    self.log.append((_name, a, k, _orig))
                return _orig(*a, **k)

            setattr(hip, name, wrapped)


def op_cost(name, a, k=None):
    """(family, algorithmic bytes, algorithmic flops) of one traced call --
    SURVEY.md 8(d): reference out-of-place semantics, fp32.  In-place scatters
    (`out=` given) are accounted with the cache-preserving minimum (tile bytes only)
    under their own family name, never mixed with the out-of-place figure."""
    e = 4
    k = k or {}
    if name in ("gather_cl", "scatter_gather_cl"):
        name = name[:-3]
    if name == "block_conv_cl":
        name = "block_conv"
    if name == "scatter_cl":
        x, y, idx = a[0], a[1], a[4]
        res = a[6] if len(a) > 6 else k.get("residual")
        inplace = (a[7] if len(a) > 7 else k.get("out")) is not None
        tiles = e * y.shape[0] * idx.shape[0] * y.shape[1] * x.shape[2] * x.shape[3] * (2 + (1 if res is not None else 0))
        return ("scatter_inplace", tiles, 0) if inplace else ("scatter", 2 * e * y.numel() + tiles, 0)
    if name == "scatter_with_block_residual_cl":
        x0, y0, x1, i0, i1 = a[0], a[1], a[2], a[6], a[8]
        inplace = (a[10] if len(a) > 10 else k.get("out")) is not None
        B, C = y0.shape[:2]
        tiles = 3 * e * B * i0.shape[0] * C * x0.shape[2] * x0.shape[3] + 4 * e * B * i1.shape[0] * C * x1.shape[2] * x1.shape[3]
        return ("scatter_block_residual_inplace", tiles, 0) if inplace else ("scatter_block_residual", 2 * e * y0.numel() + tiles, 0)
    if name in ("gather_conv_cl", "gather_conv_nchw"):
        x, x2, block, idx = a[0], a[1], a[2], a[3]
        cout, kernel, stride = a[9], a[10], a[11]
        T, cin = x.shape[0] * idx.shape[0], x.shape[1] + (0 if x2 is None else x2.shape[1])
        ro, so = (block[0] - kernel[0]) // stride[0] + 1, (block[1] - kernel[1]) // stride[1] + 1
        dense = True
        if name == "gather_conv_cl":
            full = k.get("full")
            # dense layer = every output pixel covered by a tile; a sparse tile list written into a full tensor is
            # the conv -> Scatter fusion of a SIGE layer
            dense = full is not None and idx.shape[0] * ro * so >= full["out_res"][0] * full["out_res"][1]
        return ("dense_conv_mfma" if dense else "block_conv_mfma"), 0, 2 * T * ro * so * cout * cin * kernel[0] * kernel[1]
    if name == "scatter_gather_conv_cl":
        name = "scatter_gather_conv"
    if name == "scatter_gather_conv_scatter_cl":
        y, block, idx, cout, kernel = a[1], a[2], a[3], a[10], a[11]
        T, cin = y.shape[0] * idx.shape[0], y.shape[1]
        ro, so = block[0] - kernel[0] + 1, block[1] - kernel[1] + 1
        return "block_conv_mfma", 0, 2 * T * ro * so * cout * cin * kernel[0] * kernel[1]
    if name == "gather":
        x, bH, bW, idx = a[0], a[1], a[2], a[3]
        B, C = x.shape[:2]
        return "gather", 2 * e * B * idx.shape[0] * C * bH * bW, 0
    if name == "scatter_gather":
        x, y, bH, bW, idx = a[0], a[1], a[2], a[3], a[4]
        B, C = y.shape[:2]
        n = idx.shape[0]
        return "scatter_gather", 2 * e * B * n * C * bH * bW + 12 * n * bH * bW, 0
    if name == "scatter_fused":
        x, y, n = a[0], a[1], a[3]
        res = a[4] if len(a) > 4 else None
        full = 2 * e * y.numel()
        o2 = x.shape[2] * x.shape[3]
        return "scatter", full + e * y.shape[0] * n * y.shape[1] * o2 * (2 + (1 if res is not None else 0)), 0
    if name == "scatter_with_block_residual_fused":
        x0, y0, x1, n0, n1 = a[0], a[1], a[2], a[5], a[7]
        B, C = y0.shape[:2]
        return ("scatter_block_residual",
                2 * e * y0.numel() + 3 * e * B * n0 * C * x0.shape[2] * x0.shape[3]
                + 4 * e * B * n1 * C * x1.shape[2] * x1.shape[3], 0)
    if name == "block_conv":
        x, cout, kernel, stride = a[0], a[3], a[4], a[5]
        T, cin, R, S = x.shape
        ro, so = (R - kernel[0]) // stride[0] + 1, (S - kernel[1]) // stride[1] + 1
        return "block_conv_mfma", 0, 2 * T * ro * so * cout * cin * kernel[0] * kernel[1]
    if name in ("gather_conv", "scatter_gather_conv"):
        # fused producer + conv: MFMA-bound; flops of the conv only (the gather adds no flops worth counting)
        if name == "gather_conv":
            x, block, idx, cout, kernel, stride = a[0], a[1], a[2], a[8], a[9], a[10]
            T, cin = x.shape[0] * idx.shape[0], x.shape[1]
        else:
            y, block, idx, cout, kernel, stride = a[1], a[2], a[3], a[10], a[11], a[12]
            T, cin = y.shape[0] * idx.shape[0], y.shape[1]
        ro, so = (block[0] - kernel[0]) // stride[0] + 1, (block[1] - kernel[1]) // stride[1] + 1
        return "block_conv_mfma", 0, 2 * T * ro * so * cout * cin * kernel[0] * kernel[1]
    if name == "block_conv_direct":
        x, w, stride, groups = a[0], a[1], a[3], a[4]
        T, cin, R, S = x.shape
        cout, cig, kh, kw = w.shape
        ro, so = (R - kh) // stride[0] + 1, (S - kw) // stride[1] + 1
        return "block_conv_direct", 0, 2 * T * ro * so * cout * cig * kh * kw
    raise KeyError(name)


def a_numel(t):
    return t.numel()


def pmc_traffic(family):
    """HBM-side bytes per launch of a kernel family from the committed rocprofv3 PMC pass (bench.py cannot
    run the profiler on itself); None if the file or the family is missing."""
    try:
        with open(os.path.join(REPO, "profiles", "r1k_pmc_traffic.json")) as f:
            fam = json.load(f)["families"][family]
        return int(fam["traffic_MB_per_launch"] * 1e6)
    except Exception:
        return None


def shape_key(name, a):
    parts = [name]
    for v in a:
        if isinstance(v, torch.Tensor):
            parts.append(tuple(v.shape))
        elif isinstance(v, (int, str, bool, tuple)) or v is None:
            parts.append(v)
    return tuple(parts)


def time_graph_of(fn, reps, iters=5):
    """Average device time of one `fn()` launch: a hipGraph of `reps` back-to-back
    launches, replayed `iters` times between HIP events on the capture stream."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(iters):
            g.replay()
        b.record(s)
        s.synchronize()
    torch.cuda.current_stream().wait_stream(s)
    return a.elapsed_time(b) * 1e3 / (reps * iters)  # us per launch


def capture(model, x, t):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model(x, t)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = model(x, t)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    return g, out


def timed_replays(g, steps, warmup, world):
    for _ in range(warmup):
        g.replay()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        v = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        dt = float(v.item())
    return dt


def eager_ms(model, x, t, steps):
    for _ in range(3):
        model(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model(x, t)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


# ---------------------------------------------------------------- cpu baseline --
def cpu_baseline(cfg_ratio, seconds):
    """The same U-Net + masks on the host cores, native ops from oracle/_ref (the
    reference's own compiled sige/cpu backend) or, if that is absent, from the C
    restatement.  Bounded to ~`seconds` of sparse forwards."""
    from oracle import build_ref, oracle
    from sige_amd import runtime
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    ref = None
    try:
        ref = build_ref.load()
    except Exception:
        ref = None
    kind = "reference" if ref is not None else "port"
    # threads actually used: the affinity mask, capped -- OpenMP over 256 hardware
    # threads on loops this small is slower than 16-32 (measured on the GPU box)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("SIGE_CPU_THREADS", "32"))))
    torch.set_num_threads(cores)
    oracle.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    runtime.register_backend("cpu", ref if ref is not None else oracle)
    try:
        torch.manual_seed(0)
        model = DDPMSparseUNet(DDPMConfig()).eval()
        x0 = torch.randn(1, 3, 256, 256)
        mask = square_mask(cfg_ratio)
        x1 = x0 + torch.randn(1, 3, 256, 256) * mask
        t = torch.zeros(1)
        with torch.no_grad():
            model.set_mode("full")
            model(x0, t)
            model.set_masks(downsample_mask(dilate_mask(mask, 5), 8))
            model.set_mode("sparse")
            model(x1, t)  # warm-up
            n, t0 = 0, time.perf_counter()
            while True:
                model(x1, t)
                n += 1
                dt = time.perf_counter() - t0
                if dt >= seconds or n >= 200:
                    break
    finally:
        runtime.unregister_backend("cpu")
    return {"value": round(n / dt, 3), "unit": "forward/s", "ms_per_forward": round(dt / n * 1e3, 2), "cores": cores,
            "kind": kind,
            "sample": "%d sparse DDPM-256 U-Net forwards at %.1f%% edit (same model/masks as the GPU run, "
                      "native ops = %s, convs = torch CPU)" % (n, cfg_ratio * 100,
                                                             "oracle/_ref (reference sige/cpu)" if ref is not None
                                                             else "oracle C restatement")}


# ------------------------------------------------------------------------ main --
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ratio", type=float, default=0.012, help="edit ratio of the headline workload")
    ap.add_argument("--sweep", default="0.012,0.05,0.15", help="edit ratios for the sweep section ('' = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--layout", default="nhwc", choices=["nhwc", "nchw"],
                    help="memory format of the activations: nhwc = torch.channels_last (default), nchw = the reference's")
    ap.add_argument("--no-inplace-scatter", action="store_true",
                    help="Scatter modules return a fresh full tensor per call (reference semantics) instead of "
                         "updating a persistent output buffer")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path; see DESIGN.md)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N"

    from sige_amd import hip, parallel
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    hip.lib()
    tracer = Tracer(hip)
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).to(dev).eval()  # random init: no checkpoints offline
    n_params = sum(p.numel() for p in model.parameters())
    gen = torch.Generator(device="cpu").manual_seed(1)
    x0 = torch.randn(1, 3, 256, 256, generator=gen).to(dev)
    noise = torch.randn(1, 3, 256, 256, generator=gen).to(dev)
    if args.layout == "nhwc":
        model = model.to(memory_format=torch.channels_last)
        x0, noise = x0.contiguous(memory_format=torch.channels_last), noise.contiguous(memory_format=torch.channels_last)
    model.set_scatter_inplace(args.layout == "nhwc" and not args.no_inplace_scatter)
    t = torch.zeros(1, device=dev)

    def edited(ratio):
        # each rank edits its own region of the shared original
        m = square_mask(ratio, top=100 - 6 * rank, left=90 + 6 * rank).to(dev)
        return m, x0 + noise * m

    result = {}
    with torch.no_grad():
        # ---- dense baseline: the stock U-Net forward on the same GPU ------------
        model.set_mode("full")
        model.set_plain_dense(True)
        mask, x1 = edited(args.ratio)
        gd, _ = capture(model, x1, t)
        dense_ms = timed_replays(gd, max(10, args.steps // 10), 3, 1) * 1e3 / max(10, args.steps // 10)
        del gd
        dense_by_layout = {args.layout: round(dense_ms, 3)}
        if args.layout == "nhwc":
            # the stock model in the reference's own layout too; the baseline is the faster of the two
            model.to(memory_format=torch.contiguous_format)
            gd, _ = capture(model, x1.contiguous(), t)
            d2 = timed_replays(gd, max(10, args.steps // 10), 3, 1) * 1e3 / max(10, args.steps // 10)
            del gd
            model.to(memory_format=torch.channels_last)
            dense_by_layout["nchw"] = round(d2, 3)
            dense_ms = min(dense_ms, d2)
        result["dense_forward_ms_by_layout"] = dense_by_layout
        model.set_plain_dense(False)

        # ---- cache of the original image: rank 0 computes, RCCL broadcast --------
        model(x0, t)  # every rank runs it once so shapes / cache slots exist
        flat = parallel.pack_caches(model)  # every cache tensor is now a view of `flat`
        n_cached = len(parallel.cache_slots(model))
        torch.cuda.synchronize()
        bcast_ms = 0.0
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            parallel.broadcast_cache(flat, src=0)
            torch.cuda.synchronize()
            bcast_ms = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            parallel.broadcast_cache(flat, src=0)  # second = steady-state (communicator warm)
            torch.cuda.synchronize()
            bcast_ms = min(bcast_ms, (time.perf_counter() - t0) * 1e3)

        def prepare(ratio):
            m, xe = edited(ratio)
            model.set_masks(downsample_mask(dilate_mask(m, 5), 8))  # diffusion/runner.py:157-165
            model.set_mode("sparse")
            return xe

        # ---- headline: sparse forward at --ratio ----------------------------------
        x1 = prepare(args.ratio)
        model(x1, t)  # packs weights, builds tile tables
        tracer.log = []
        model(x1, t)
        trace, tracer.log = tracer.log, None
        e_ms = eager_ms(model, x1, t, 20)
        g, out = capture(model, x1, t)
        dt = timed_replays(g, args.steps, args.warmup, world)
        ms_per_step = dt * 1e3 / args.steps
        assert torch.isfinite(out).all()

        if rank == 0:
            # ---- per-kernel accounting of the hot path (warm, in-situ tensors) ----
            fam = {}
            per_cfg = {}
            for name, a, k, orig in trace:
                key = shape_key(name, a)
                if key not in per_cfg:
                    family, nbytes, flops = op_cost(name, a, k)
                    us = time_graph_of(lambda: orig(*a, **k), reps=8)
                    per_cfg[key] = dict(family=family, bytes=nbytes, flops=flops, us=us, count=0, call=(orig, a, k))
                per_cfg[key]["count"] += 1
            for c in per_cfg.values():
                f = fam.setdefault(c["family"], dict(us=0.0, bytes=0, flops=0, launches=0))
                f["us"] += c["us"] * c["count"]
                f["bytes"] += c["bytes"] * c["count"]
                f["flops"] += c["flops"] * c["count"]
                f["launches"] += c["count"]
            kernels = {}
            for name, f in fam.items():
                kernels[name] = {"launches": f["launches"], "us_total": round(f["us"], 1)}
                if f["bytes"]:
                    kernels[name]["alg_MB"] = round(f["bytes"] / 1e6, 1)
                    kernels[name]["alg_GBps"] = round(f["bytes"] / f["us"] / 1e3, 1)
                if f["flops"]:
                    kernels[name]["GFLOP"] = round(f["flops"] / 1e9, 2)
                    kernels[name]["TFLOPs"] = round(f["flops"] / f["us"] / 1e6, 2)
            conv = [f for n, f in fam.items() if n.startswith("block_conv")]  # the active-block (SIGE) convs
            conv_tflops = sum(f["flops"] for f in conv) / max(1e-9, sum(f["us"] for f in conv)) / 1e6
            hot_us = sum(f["us"] for f in fam.values())
            result.update(kernels=kernels, hot_path_us=round(hot_us, 1), block_conv_tflops=round(conv_tflops, 2))

            # ---- roofline of the dominant hot-path kernel family ---------------------
            if not args.no_roofline:
                dom = max((n for n in fam if n != "dense_conv_mfma"), key=lambda n: fam[n]["us"])  # hot path = the SIGE ops
                cfgs = [c for c in per_cfg.values() if c["family"] == dom]
                # cold measurement: clone the inputs into >= 6 rotating sets so that the
                # 256 MiB Infinity Cache does not serve them
                tot_us, tot_work, launches = 0.0, 0.0, 0
                for c in cfgs:
                    orig, a, k = c["call"]
                    nbytes_in = sum(v.numel() * v.element_size() for v in a if isinstance(v, torch.Tensor))
                    nsets = max(2, min(8, int(600e6 // max(1, nbytes_in)) + 1))
                    sets = [tuple(v.clone() if isinstance(v, torch.Tensor) and v.is_floating_point() and v.numel() > 4096
                                  else v for v in a) for _ in range(nsets)]
                    it = [0]

                    def rot():
                        orig(*sets[it[0] % nsets], **k)
                        it[0] += 1

                    us = time_graph_of(rot, reps=nsets * 2)
                    tot_us += us * c["count"]
                    tot_work += (c["flops"] if c["flops"] else c["bytes"]) * c["count"]
                    launches += c["count"]
                    del sets
                is_mfma = fam[dom]["flops"] > 0
                achieved = tot_work / tot_us / (1e6 if is_mfma else 1e3)
                peak = PEAK_F32_MFMA_TFS if is_mfma else PEAK_HBM_GBS
                result["roofline"] = {
                    "kernel": dom, "bound": "mfma" if is_mfma else "hbm",
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s" if is_mfma else "GB/s",
                    "frac": round(achieved / peak, 4), "traffic": pmc_traffic(dom),
                    "launches_per_forward": launches, "avg_launch_us": round(tot_us / launches, 2),
                    "work_per_launch": round(tot_work / launches / (1e9 if is_mfma else 1e6), 4),
                    "work_unit": "GFLOP" if is_mfma else "MB",
                    "note": "algorithmic work of all %d launches of this kernel in one forward / their summed "
                            "duration (hipGraph of back-to-back launches, HIP events on the launch stream, "
                            "rotating input sets); traffic = bytes per launch from the committed PMC pass "
                            "profiles/r1k_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs, "
                            "FETCH_SIZE doubled per the gfx950 correction; L2-miss traffic incl. Infinity-Cache hits)"
                            % launches}
                # secondary: the HBM-bound copy-through scatter (reference semantics: a fresh full tensor per call,
                # sige/cpu/scatter.cpp:83) at its largest shape in this model.  The forward itself no longer launches
                # it (conv -> scatter is fused into the conv's epilogue over a persistent output), so it is measured
                # standalone on the module's own cache, index list and tile table, with rotating buffers.
                from sige_amd.nn import Scatter as _Scatter

                cands = [m for m in model.modules() if isinstance(m, _Scatter) and m.original_outputs]
                if cands:
                    m = max(cands, key=lambda m: next(iter(m.original_outputs.values())).numel())
                    gm = m.gather.module
                    y = next(iter(m.original_outputs.values()))
                    idx = gm.indices_on(dev)
                    table = gm.tile_table(y.shape[2:], dev)
                    nsets = 6
                    fmt = torch.channels_last if hip.is_cl(y) else torch.contiguous_format
                    ys = [torch.randn_like(y) for _ in range(nsets)]
                    rs = [torch.randn_like(y) for _ in range(nsets)]
                    xs = [torch.randn(y.shape[0] * idx.shape[0], y.shape[1], *gm.out_tile, device=dev).contiguous(memory_format=fmt)
                          for _ in range(nsets)]
                    it = [0]

                    def rot2():
                        i = it[0] % nsets
                        if hip.is_cl(y):
                            hip.scatter_cl(xs[i], ys[i], gm.offset, gm.model_stride, idx, table, rs[i])
                        else:
                            hip.scatter_fused(xs[i], ys[i], table, idx.shape[0], rs[i])
                        it[0] += 1

                    us = time_graph_of(rot2, reps=nsets * 2)
                    nbytes = 2 * 4 * y.numel() + 4 * xs[0].numel() * 3
                    result["roofline_hbm"] = {"kernel": "scatter (out-of-place, one pass, + residual)", "shape": str(tuple(y.shape)),
                                              "layout": "nhwc" if hip.is_cl(y) else "nchw", "active_tiles": int(idx.shape[0]),
                                              "alg_MB": round(nbytes / 1e6, 1), "us": round(us, 2),
                                              "achieved": round(nbytes / us / 1e3, 1), "peak": PEAK_HBM_GBS,
                                              "unit": "GB/s", "frac": round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4)}
                    del ys, rs, xs
        del g, trace

        # ---- edit-ratio sweep (rank 0 only, short) -----------------------------------
        sweep = []
        if rank == 0 and args.sweep and world == 1:  # (at N > 1 the other ranks would idle in the final barrier)
            for r in [float(v) for v in args.sweep.split(",")]:
                xs = prepare(r)
                model(xs, t)
                tracer.log = []
                model(xs, t)
                tr, tracer.log = tracer.log, None
                flops = sum(op_cost(n, a, k)[2] for n, a, k, o in tr if op_cost(n, a, k)[0] != "dense_conv_mfma")
                gs, _ = capture(model, xs, t)
                k = max(20, args.steps // 4)
                ms = timed_replays(gs, k, 5, 1) * 1e3 / k
                n256 = max([a[3].shape[0] for n, a, kk, o in tr if n in ("gather", "gather_cl", "gather_conv_cl") and a[0].shape[2] == 256]
                           + [a[2].shape[0] for n, a, kk, o in tr if n == "gather_conv" and a[0].shape[2] == 256] + [0])
                sweep.append({"edit_ratio": r, "forward_ms": round(ms, 3), "speedup_vs_dense": round(dense_ms / ms, 2),
                              "active_tiles_256": n256, "block_conv_GFLOP": round(flops / 1e9, 2)})
                del gs, tr

    if world > 1:
        dist.barrier()
    if rank == 0:
        cpu = None
        if args.cpu_seconds > 0 and world == 1:
            cpu = cpu_baseline(args.ratio, args.cpu_seconds)
        line = {
            "metric": "DDPM-256 U-Net sparse (SIGE) forwards/s",
            "value": round(world * args.steps / dt, 2),
            "unit": "forward/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DDPM 256x256 church U-Net (ch128, mult 1-1-2-2-4-4, %.1fM params, random init), "
                                   "%.1f%% square edit, one edited image per GPU, hipGraph replay, %s activations%s"
                                   % (n_params / 1e6, args.ratio * 100, args.layout.upper(),
                                      ", in-place scatter buffers" if (args.layout == "nhwc" and not args.no_inplace_scatter) else ""),
                       "edit_ratio": args.ratio, "batch_per_gpu": 1, "resolution": 256,
                       "parallelism": "dp%d" % world},
            "forward_ms": round(ms_per_step, 4),
            "forward_ms_eager": round(e_ms, 3),
            "dense_forward_ms": round(dense_ms, 3),
            "speedup_vs_dense": round(dense_ms / ms_per_step, 2),
            "cache_bytes": int(flat.numel() * 4), "cached_tensors": n_cached,
            "cache_broadcast_ms": round(bcast_ms, 3),
            "sweep": sweep,
        }
        line.update(result)
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
