"""Shared pieces of bench.py's sections: peaks, the synthetic masks, hipGraph capture / replay timing (the reference's protocol:
warm-up, synchronize, K timed iterations, synchronize -- /root/reference/diffusion/runner.py:224-231 -- with a barrier and the
max over ranks added for N > 1)."""
import time

import torch
import torch.distributed as dist

# hipGraph captures: "thread_local" -- another thread of the process (RCCL's watchdog at N > 1) calling into HIP while this
# thread records must not invalidate the capture; the recording thread itself makes no capture-unsafe call either way.
CAPTURE_MODE = "thread_local"

PEAK_HBM_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFS = 2500.0  # v_mfma_f32_32x32x16_f16 dense peak (not the 2:1-sparsity figure)


def square_mask(ratio, H=256, W=256, top=100, left=90):
    side = int(round((ratio ** 0.5) * H))
    m = torch.zeros(H, W, dtype=torch.bool)
    m[top:top + side, left:left + side] = True
    return m


def time_graph_of(fn, reps, iters=5):
    """Average device time of one `fn()` launch: a hipGraph of `reps` back-to-back
    launches, replayed `iters` times between HIP events on the capture stream."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode=CAPTURE_MODE):
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
        with torch.cuda.graph(g, stream=s, capture_error_mode=CAPTURE_MODE):
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


def _replay_ms(fn, k=30, warm=5):
    """ms per call of `fn()` replayed as a hipGraph."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            out = fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode=CAPTURE_MODE):
            out = fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    ms = timed_replays(g, k, warm, 1) * 1e3 / k
    return ms, out, g


def _hip():
    from sige_amd import hip

    return hip


def capture_fn(fn, warm=2):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode=CAPTURE_MODE):
            out = fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    return g, out
