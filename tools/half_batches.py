"""Would two row-halves of ONE cfg-2 batch, run as two concurrent chains, beat the single chain?  Proxy with what exists: two
model instances at B = 8 (own workspaces, one hipGraph each) replayed together on two streams, against one B = 16 graph.
(The cluster op's split chunks differ between B = 8 and B = 16, so this is a TIMING probe only.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, bench_side

dev = torch.device("cuda", 0)
c = bench.CFG2
sd = bench.random_state_dict(c, 0)


def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for parts in (1, 2, 4):
    cc = dict(c, B=c["B"] // parts)
    keep, graphs, streams = [], [], []
    for s_ in range(parts):
        k, g, st = bench_side.two_in_flight_graphs(cc, sd, dev)
        keep.append(k); graphs.append(g[0]); streams.append(st[0])

    def go():
        for s_ in range(parts):
            with torch.cuda.stream(streams[s_]):
                graphs[s_].replay()
    ms = min(timed(go) for _ in range(3))
    print("%d chain(s) of B = %d in flight: %.3f ms per 16 clips" % (parts, cc["B"], ms), flush=True)
    del keep, graphs, streams
    torch.cuda.empty_cache()
