"""Where does the step time go? time the step with pieces disabled (dev tool)."""
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
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("pair (visual+text)   %.3f ms" % t(lambda: model.clip.encode_pair(v, ids, 12)))
print("visual.encode only   %.3f ms" % t(lambda: model.clip.visual.encode(v, 12)))
print("encode_text only     %.3f ms" % t(lambda: model.clip.encode_text(ids)))
