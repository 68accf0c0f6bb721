"""Dev tool: time the patch gather for fp32 / uint8 CHW / uint8 HWC input (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerclip_amd.clip4clip import CLIP4Clip
c = bench.CFG2
dev = torch.device("cuda", 0)
sd = bench.random_state_dict(c, 0)
model = CLIP4Clip.from_state_dict(dict(sd), bench.task_config(c)).to(dev).eval()
f32 = torch.randn(192, 3, 224, 224, device=dev)
chw = torch.randint(0, 256, (192, 3, 224, 224), dtype=torch.uint8, device=dev)
hwc = torch.randint(0, 256, (192, 224, 224, 3), dtype=torch.uint8, device=dev)
for name, x in (("f32", f32), ("u8 chw", chw), ("u8 hwc", hwc)):
    ms = bench.event_time_ms(lambda: model.clip.visual.encode(x, 12), 10)
    print(name, "visual.encode %.3f ms" % ms)
