"""Cost of a cross-stream event hop between hipGraph replays on this box: main replays A, records e1; side waits e1 and records
e2 (no work); main replays B, then waits e2 before replaying C.  Compared with the same three replays without the hop."""
import time
import torch

dev = torch.device("cuda:0")
big = torch.zeros(8 << 20, device=dev, dtype=torch.bfloat16)


def make_graph(n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        big.add_(1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            big.add_(1.0)
    return g


A, B, C = make_graph(300), make_graph(300), make_graph(300)
side = torch.cuda.Stream()
e1, e2 = torch.cuda.Event(), torch.cuda.Event()


def plain():
    A.replay(); B.replay(); C.replay()


def hop():
    main = torch.cuda.current_stream()
    A.replay()
    e1.record(main)
    with torch.cuda.stream(side):
        side.wait_event(e1)
        e2.record(side)
    B.replay()
    main.wait_event(e2)
    C.replay()


def hop_same_stream():
    main = torch.cuda.current_stream()
    A.replay()
    e1.record(main)
    main.wait_event(e1)
    e2.record(main)
    B.replay()
    main.wait_event(e2)
    C.replay()


for name, fn in (("plain", plain), ("cross-stream hop", hop), ("same-stream events", hop_same_stream), ("plain", plain)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name:22s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per iteration (3 graphs x 300 nodes)")
