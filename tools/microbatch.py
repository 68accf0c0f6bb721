"""Experiment: one 16-clip step vs two 8-clip micro-batches on two streams (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerclip_amd.clip4clip import CLIP4Clip
c = bench.CFG2
dev = torch.device("cuda", 0)
sd = bench.random_state_dict(c, 0)
model = CLIP4Clip.from_state_dict(dict(sd), bench.task_config(c)).to(dev).eval()
ids, amask, video, vmask = bench.synthetic_batch(c, dev, 100)
v = video.view(-1, 3, 224, 224)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("B=16 one stream   %.3f ms" % t(lambda: model.clip.encode_pair(v, ids, 12)))
for nsplit in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    per = 16 // nsplit
    vs = [v[i * per * 12:(i + 1) * per * 12].contiguous() for i in range(nsplit)]
    idss = [ids[i * per:(i + 1) * per].contiguous() for i in range(nsplit)]
    def run():
        cur = torch.cuda.current_stream()
        for s_, vv, ii in zip(streams, vs, idss):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                model.clip.encode_pair(vv, ii, 12)
        for s_ in streams:
            cur.wait_stream(s_)
    print("B=16 as %d x %d on %d streams  %.3f ms" % (nsplit, per, nsplit, t(run)))
    g = torch.cuda.CUDAGraph()
    run(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        run()
    print("   ... hipGraph replay      %.3f ms" % t(g.replay))
