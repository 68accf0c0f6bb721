"""bench_common.py - what bench.py and its side measurements (bench_side.py) share: the BASELINE.json configurations, the
synthetic weights / batches and the two timing helpers (HIP events on the launch stream; hipGraph of launches)."""
import os
import sys
from argparse import Namespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_F16_PEAK_TFLOPS = 2500.0       # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0               # HBM3E spec peak

CFG2 = dict(name="cfg2 MSR-VTT-shaped: ViT-B/32 224^2, 12 frames -> 3 segments @block 7, K=49, batch 16, 32 words",
            B=16, T=12, T_new=3, K=49, cluster_block=7, words=32, patch=32, res=224, width=768, layers=12)
# The towers of the other BASELINE.json configs at their per-GPU batch: timed by forward_config_bench()
# next to the headline and selectable as the timed workload with --workload (that is how profiles/r05_forward_cfg* were taken)
FORWARD_CFGS = {
    "cfg2": CFG2,
    "cfg3": dict(name="cfg3 MSVD-shaped (per GPU): ViT-B/32 224^2, 12 frames -> 4 segments @block 7, K=49, batch 64, 32 words",
                 B=64, T=12, T_new=4, K=49, cluster_block=7, words=32, patch=32, res=224, width=768, layers=12),
    "cfg4": dict(name="cfg4 ActivityNet-shaped (per GPU): ViT-B/32 224^2, 64 frames -> 8 segments @block 7, K=49, batch 8, 77 words",
                 B=8, T=64, T_new=8, K=49, cluster_block=7, words=77, patch=32, res=224, width=768, layers=12),
    "cfg5": dict(name="cfg5 ViT-B/16 224^2 (per GPU): 12 frames -> 4 segments @block 7, 196 tokens/frame, K=100, split 4, batch 16, 32 words",
                 B=16, T=12, T_new=4, K=100, cluster_block=7, words=32, patch=16, res=224, width=768, layers=12, split=4),
}
# cluster-op shapes of the other BASELINE.json configs: reported as µs/call + Mtokens/s
CLUSTER_SHAPES = {"cfg2": dict(B=16, T=12, T_new=3, n=49, K=49, split=16),
                  "cfg3 MSVD-shaped (per GPU)": dict(B=64, T=12, T_new=4, n=49, K=49, split=16),      # P = 256 problems: fills the chip
                  "cfg4 ActivityNet-shaped (per GPU)": dict(B=8, T=64, T_new=8, n=49, K=49, split=16),
                  "cfg5 ViT-B/16": dict(B=16, T=12, T_new=4, n=196, K=100, split=4),
                  # scripts/activitynet.sh:104-122 (ViT-B/16, 60 -> 15 frames, K = 160, batch 4 per GPU): N = 784
                  "cfg6 ViT-B/16 ActivityNet (per GPU)": dict(B=4, T=60, T_new=15, n=196, K=160, split=4)}


def task_config(c):
    return Namespace(cluster_inter=1, cluster_algo='kmediods++', max_frames=c["T"],
                     target_frames_blocks=[c["T"]] * (c["cluster_block"] - 1) + [c["T_new"]] * (13 - c["cluster_block"]),
                     cluster_num_blocks=[c["K"]] * 12, cluster_distance='euclidean', cluster_threshold=1e-6,
                     cluster_iter_limit=100, minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/%d' % c["patch"], aggregation=None,
                     pre_norm=False, loose_type=True, sim_header='meanP', linear_patch='2d')


def algorithmic_flops_per_clip(c):
    """2 flops per multiply-add; the cluster op fires before the attention of block `cluster_block`; the
    projection heads count the CLS / EOT rows only."""
    W, p, T, Tn, cb = c["width"], c["patch"], c["T"], c["T_new"], c["cluster_block"]
    n = (c["res"] // p) ** 2
    L0, L1, Lt = 1 + n, 1 + c["K"], c["words"]
    f_vis = (T * n * 2 * (3 * p * p) * W + (cb - 1) * T * L0 * (24 * W * W + 4 * L0 * W)
             + (13 - cb) * Tn * L1 * (24 * W * W + 4 * L1 * W) + Tn * 2 * W * 512)
    f_txt = 12 * Lt * (24 * 512 * 512 + 4 * Lt * 512) + 2 * 512 * 512
    return float(f_vis + f_txt)


def random_state_dict(c, seed):
    """Random-init weights of the named architecture with CLIP.initialize_parameters statistics
    (modules/clip.py:419-446), rounded through fp16 as convert_weights does."""
    from centerclip_amd.clip import CLIP
    torch.manual_seed(seed)
    m = CLIP(512, c["res"], c["layers"], c["width"], c["patch"], 77, 49408, 512, 8, 12, video_frames=c["T"], args=None)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.half().float())
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def synthetic_batch(c, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    video = torch.randn(c["B"], 1, c["T"], 3, c["res"], c["res"], generator=g)
    vmask = torch.ones(c["B"], 1, c["T"], dtype=torch.long)
    vmask[-1, 0, c["T"] - 2:] = 0                      # one clip with trailing padding frames
    ids = torch.zeros(c["B"], c["words"], dtype=torch.long)
    for b in range(c["B"]):
        ln = int(torch.randint(4, c["words"] + 1, (1,), generator=g))
        ids[b, 0], ids[b, ln - 1] = 49406, 49407
        ids[b, 1:ln - 1] = torch.randint(1, 49405, (ln - 2,), generator=g)
    amask = (ids > 0).long()
    return [t.to(device) for t in (ids, amask, video, vmask)]


def event_time_ms(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def graph_time_ms(fn, launches=20, replays=4):
    """Average duration of one launch of `fn`: `launches` of them captured into one hipGraph, replayed with HIP events
    around the replays on the launch stream - the same way the step itself is issued, so the interval holds the kernels
    and the graph's own launch-to-launch gaps, not the host's eager launch cadence.  Falls back to eager launches."""
    try:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread of an initialised process group may query events meanwhile
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        with torch.cuda.graph(g, capture_error_mode=mode):
            for _ in range(launches):
                fn()
        g.replay()
        torch.cuda.synchronize()
        return event_time_ms(g.replay, replays, warm=1) / launches
    except Exception as exc:                   # noqa: BLE001
        sys.stderr.write("graph timing failed (%s); timing eager launches\n" % exc)
        torch.cuda.synchronize()
        return event_time_ms(fn, launches)
