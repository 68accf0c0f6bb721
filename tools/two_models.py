"""dev check: two CLIP4Clip instances in one process - eager on one stream, eager on two streams, captured on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerclip_amd.clip4clip import CLIP4Clip
dev = torch.device("cuda:0")
c = bench.CFG2
sd = bench.random_state_dict(c, seed=0)
stage = sys.argv[1] if len(sys.argv) > 1 else "all"
ms, steps = [], []
for s in range(2):
    m = CLIP4Clip.from_state_dict(dict(sd), bench.task_config(c)).to(dev).eval()
    ids, amask, video, vmask = bench.synthetic_batch(c, dev, seed=100)
    tt = torch.zeros_like(ids)
    def step(m=m, ids=ids, tt=tt, amask=amask, video=video, vmask=vmask):
        with torch.no_grad():
            out = m(ids, tt, amask, video, vmask)
            return m.get_similarity_logits(out["sequence_output"], out["visual_output"], amask, vmask)[0]
    ms.append(m); steps.append(step)
a = steps[0](); b = steps[1](); torch.cuda.synchronize()
print("one stream, eager: equal", bool(torch.equal(a, b)), flush=True)
if stage in ("all", "streams", "graphs"):
    sts = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for r in range(3):
        for s in range(2):
            with torch.cuda.stream(sts[s]):
                outs.append(steps[s]())
    torch.cuda.synchronize()
    print("two streams, eager: equal", all(bool(torch.equal(o, a)) for o in outs), flush=True)
if stage in ("all", "graphs"):
    graphs, gouts = [], []
    for s in range(2):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(sts[s]):
            with torch.cuda.graph(g, stream=sts[s]):
                gouts.append(steps[s]())
        torch.cuda.synchronize()
        graphs.append(g)
        print("captured", s, flush=True)
    for s in range(2):
        with torch.cuda.stream(sts[s]):
            graphs[s].replay()
        torch.cuda.synchronize()
        print("replayed alone", s, bool(torch.equal(gouts[s], a)), flush=True)
    for r in range(20):
        for s in range(2):
            with torch.cuda.stream(sts[s]):
                graphs[s].replay()
    torch.cuda.synchronize()
    print("replayed concurrently: equal", bool(torch.equal(gouts[0], a)), bool(torch.equal(gouts[1], a)), flush=True)
    def timed(fn, n):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    def one():
        with torch.cuda.stream(sts[0]): graphs[0].replay()
    def two():
        for s in range(2):
            with torch.cuda.stream(sts[s]): graphs[s].replay()
    for r in range(3):
        x = timed(one, 200); y = timed(two, 100)
        print("one stream %.3f ms per step | two streams %.3f ms per pair = %.3f ms per step" % (x, y, y / 2), flush=True)
