"""Per-node cost of a serial hipGraph replay on this box: N dependent tiny kernels (and N dependent ~20 us kernels) in one
captured graph, timed per replay.  Tells how much of a step's wall time is dispatch overhead rather than kernel time."""
import time
import torch

dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
big = torch.zeros(32 << 20, device=dev, dtype=torch.bfloat16)  # 64 MiB: one add_ ~ 30 us


def bench(fn, n, label):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # eager for comparison
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    de = time.perf_counter() - t0
    print(f"{label}: graph replay {dt * 1e6 / n:.2f} us per node, eager {de * 1e6 / n:.2f} us per launch ({n} nodes)")


bench(lambda: x.add_(1.0), 1000, "tiny kernel (64 floats)")
bench(lambda: big.add_(1.0), 200, "64 MiB bf16 add_ (128 MiB traffic)")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(50):
    big.add_(1.0)
ev1.record()
torch.cuda.synchronize()
print(f"64 MiB add_ back to back, events: {ev0.elapsed_time(ev1) * 1e3 / 50:.2f} us each -> {2 * big.numel() * 2 / (ev0.elapsed_time(ev1) * 1e-3 / 50) / 1e12:.2f} TB/s")
