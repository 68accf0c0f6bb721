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
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
full = lambda: model.get_similarity_logits(*(lambda o: (o["sequence_output"], o["visual_output"]))(model(ids, torch.zeros_like(ids), amask, video, vmask)), amask, vmask)
print("full step            %.3f ms" % t(full))
print("video only (forward) %.3f ms" % t(lambda: model(None, None, None, video, vmask)))
print("text only            %.3f ms" % t(lambda: model(ids, torch.zeros_like(ids), amask)))
v = video.view(-1, 3, 224, 224)
print("visual.encode only   %.3f ms" % t(lambda: model.clip.visual.encode(v, 12)))
print("encode_text only     %.3f ms" % t(lambda: model.clip.encode_text(ids)))
# serial (no side stream)
def serial():
    s = model.get_sequence_output(ids); vv, _ = model.get_visual_output(v, vmask.view(-1, 12)[:, [3, 7, 11]], video_frame=12)
    return s, vv
print("serial text+video    %.3f ms" % t(serial))
