"""Dev tool: n independent cfg-2 batches in flight (n model instances, one hipGraph each, n streams), resident inputs (GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, bench_side

dev = torch.device("cuda", 0)
c = bench.CFG2
sd = bench.random_state_dict(c, 0)
keep, graphs, streams = [], [], []
for s_ in range(2):
    k, g, st = bench_side.two_in_flight_graphs(c, sd, dev)
    keep.append(k); graphs += list(g); streams += list(st)


def timed(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for parts in (1, 2, 3, 4):
    def go():
        for s_ in range(parts):
            with torch.cuda.stream(streams[s_]):
                graphs[s_].replay()
    ms = min(timed(go) for _ in range(3)) / parts
    print("%d batch(es) in flight: %.3f ms per batch = %.0f clips/s" % (parts, ms, 16 / ms * 1e3), flush=True)
