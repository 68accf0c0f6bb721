"""Two batches in flight WITHOUT hipGraphs (what eval_epoch(in_flight=2) does): a replica of the model (CLIP4Clip.replica-style)
on a second stream, batches alternating; checks identical features and prints ms per 16-clip batch for 1 and 2 in flight."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from centerclip_amd.clip4clip import CLIP4Clip
dev = torch.device("cuda", 0)
c = bench.CFG2
sd = bench.random_state_dict(c, seed=0)
m0 = CLIP4Clip.from_state_dict(dict(sd), bench.task_config(c)).to(dev).eval()
batches = [bench.synthetic_batch(c, dev, seed=900 + i) for i in range(8)]
def run(m, b):
    ids, amask, video, vmask = b
    out = m(ids, torch.zeros_like(ids), amask, video, vmask)
    return out["sequence_output"].clone(), out["visual_output"].clone()
with torch.no_grad():
    ref = [run(m0, b) for b in batches]
    torch.cuda.synchronize()
    m1 = CLIP4Clip.from_state_dict({k: v.detach() for k, v in m0.clip.state_dict().items()}, m0.task_config).to(dev).eval()
    models, streams = [m0, m1], [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    outs = []
    for i, b in enumerate(batches):
        with torch.cuda.stream(streams[i % 2]):
            outs.append(run(models[i % 2], b))
    torch.cuda.synchronize()
    ok = all(torch.equal(a[0], r[0]) and torch.equal(a[1], r[1]) for a, r in zip(outs, ref))
    print("two replicas on two streams identical to sequential:", ok)
    for mode in (1, 2):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            for k in range(200):
                i = k % 8
                if mode == 1:
                    run(m0, batches[i])
                else:
                    with torch.cuda.stream(streams[k % 2]):
                        run(models[k % 2], batches[i])
            torch.cuda.synchronize(); dt = time.time() - t0
        print("in flight %d: %.3f ms per 16-clip batch (eager)" % (mode, dt / 200 * 1e3))
