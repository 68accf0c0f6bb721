"""Dev tool: eval_epoch throughput from pinned host batches (cfg 2, uint8 HWC) for in_flight x graphed (GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
import bench
from centerclip_amd.clip4clip import CLIP4Clip
from centerclip_amd import eval as ev

dev = torch.device("cuda", 0)
c = bench.CFG2
sd = bench.random_state_dict(c, 0)
model = CLIP4Clip.from_state_dict(dict(sd), bench.task_config(c)).to(dev).eval()
g = torch.Generator().manual_seed(9)
host = []
for i in range(4):
    ids, amask, video, vmask = [t.cpu() for t in bench.synthetic_batch(c, "cpu", seed=500 + i)]
    video = torch.randint(0, 256, (c["B"], 1, c["T"], 224, 224, 3), dtype=torch.uint8, generator=g)
    host.append(tuple(t.pin_memory() for t in (ids, amask, torch.zeros_like(ids), video, vmask)))


class Loader(list):
    pass


n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
loader = Loader(host[i % 4] for i in range(n))
loader.dataset = Namespace(multi_sentence_per_video=False)
args = Namespace(inference_speed_test=True)
for graphed in (False, True):
    for fl in (1, 2, 3, 4):
        kw = dict(in_flight=fl, graphed=graphed)
        ev.eval_epoch(model, loader, dev, args=args, **kw)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ev.eval_epoch(model, loader, dev, args=args, **kw)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("graphed=%d in_flight=%d: %.3f ms per batch, %.0f clips/s" % (graphed, fl, best / n * 1e3, n * 16 / best), flush=True)
