#!/usr/bin/env python
"""bench.py - headline benchmark of the CenterCLIP retrieval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this script launches itself under torch.distributed.run with N ranks (one
per GPU, RCCL) and refuses - exit code 2 - when fewer than N GPUs are visible.  Launched by torch.distributed.run it
checks WORLD_SIZE == N.

Metric (BASELINE.json): clips/sec (ViT-B/32, 12 frames) + token-cluster Mtokens/s, and pairwise-similarities/s as an
extra field.  Workload = BASELINE.json configs[1] ("MSR-VTT-shaped synthetic: ViT-B/32, 12 frames, 3 segments, k=49
medoids, batch 16") per GPU.

One step = one pass of the hot path over one batch that is already resident in HBM:
  video [16,1,12,3,224,224] fp32 + ids [16,32]  ->  CLIP4Clip.forward (text tower, ViT with the token-cluster op in
  block 7)  ->  [N>1: ONE RCCL all-gather of the preallocated feature records]  ->  get_similarity_logits (this rank's
  row block of the [G*16, G*16] logits).
Weak scaling: every rank owns its own 16 clips; value = all ranks' clips / max-over-ranks time.

Timing: W untimed warm-up steps, then windows of EXACTLY K steps, each bracketed by barrier + synchronize on both
sides, repeated until >= --min-seconds of timed work have run (DVFS steady state; the first window alone is ~40 ms).
`ms_per_step` / `value` are the MEDIAN window (max over ranks per window); the first window is reported next to it.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the kernel symbol with the
largest share of the step (measured live with HIP events on the launch stream) and `cpu_baseline` (the oracle = the
plain-PyTorch CPU restatement of the reference path, timed on this host on a bounded sample).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.abspath(__file__))
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


TILES = {1: (128, 128, 2, 2, 64), 2: (128, 64, 2, 2, 64), 3: (64, 128, 2, 2, 64), 4: (64, 64, 2, 2, 64),
         5: (256, 256, 2, 4, 64), 6: (256, 128, 4, 2, 64), 7: (256, 192, 2, 4, 64), 8: (64, 64, 2, 2, 128), 10: (128, 256, 2, 4, 64)}


def kernel_symbol(M, N, K, epi):
    """Name of the gemm_f16_kernel instantiation a stand-alone launch of this shape runs on (as rocprofv3 prints it)."""
    from centerclip_amd import _lib as L
    if epi == 8:                                 # in_proj + attention in one launch: always the 256x192 tile
        return "gemm_f16_kernel<256, 192, 2, 4, 8, 64>"
    t = L.lib().cc_linear_tile_for(M, N, K, epi)
    bm, bn, wm, wn, bk = TILES[t]
    return "gemm_f16_kernel<%d, %d, %d, %d, %d, %d>" % (bm, bn, wm, wn, epi, bk)


PMC_PASSES = {   # one counter group per rocprofv3 run (4 TCC slots; counters + kernel trace only)
    "read": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
    "write": ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
}
_PMC_CACHE = {}


def pmc_counters_live():
    """The L2 <-> fabric request counters of every kernel INSIDE this invocation's step: bench.py re-runs its own step (3 eager
    steps of the same workload, no side measurements) under `rocprofv3 --pmc <group> --kernel-trace`, one counter group per
    child run, and averages the counters per kernel symbol.  -> {kernel name: {counter: average per launch, "launches": n}} or
    {"error": reason}.  Calibration of the byte arithmetic (profiles/r05_traffic_reconcile.txt): a 512 MiB device copy reads
    4,194,510 requests = 128 B each (TCC_EA0_RDREQ_128B is not populated on gfx950; RDREQ - 32B - 64B are the 128-byte ones) and
    writes 8,388,608 requests of 64 B; FETCH_SIZE of the same launch reads exactly half of the bytes - the guide's x2 rule, which
    also holds for the GEMM's LDS-DMA loads (one-column-tile launch: 59 MB A + 8 x 0.79 MB W expected, 65.4 MB counted)."""
    if "data" in _PMC_CACHE:
        return _PMC_CACHE["data"]
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        _PMC_CACHE["data"] = {"error": "rocprofv3 not found"}
        return _PMC_CACHE["data"]
    out = {}
    base = tempfile.mkdtemp(prefix="cc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("WORLD_SIZE", None)
    try:
        for name, counters in PMC_PASSES.items():
            d = os.path.join(base, name)
            cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "-d", d, "-o", name, "--output-format", "csv", "--", sys.executable,
                                               os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--min-seconds", "0",
                                               "--no-extras", "--no-cpu-baseline", "--no-graph"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                _PMC_CACHE["data"] = {"error": "rocprofv3 pass '%s' failed (rc %d): %s" % (name, r.returncode, r.stderr.decode(errors="replace")[-300:])}
                return _PMC_CACHE["data"]
            acc = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    acc.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for k, cs in acc.items():
                e = out.setdefault(k, {})
                for c_, v in cs.items():
                    e[c_] = sum(v) / len(v)
                    e["launches"] = len(v)
    except Exception as exc:                     # noqa: BLE001
        _PMC_CACHE["data"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        return _PMC_CACHE["data"]
    finally:
        shutil.rmtree(base, ignore_errors=True)
    _PMC_CACHE["data"] = out
    return out


def pmc_bytes(c):
    """Bytes across the L2 <-> fabric interface from one kernel's averaged request counters (see pmc_counters_live)."""
    rd, r32, r64 = c.get("TCC_EA0_RDREQ_sum", 0.0), c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_64B_sum", 0.0)
    wr, w64 = c.get("TCC_EA0_WRREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    fetch = 128.0 * (rd - r32 - r64) + 64.0 * r64 + 32.0 * r32
    write = 64.0 * w64 + 32.0 * (wr - w64)
    return fetch, write


def pmc_traffic(symbol):
    """Fabric bytes per launch of a kernel symbol, measured in THIS invocation (pmc_counters_live); None + the reason when the
    counter passes could not run."""
    data = pmc_counters_live()
    if "error" in data:
        return {"hbm_bytes_per_launch": None, "source": "not measured: " + data["error"]}
    key = symbol.replace(" ", "")
    for name, v in data.items():
        if key in name.replace(" ", ""):
            fetch, write = pmc_bytes(v)
            return {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes": round(fetch), "write_bytes": round(write),
                    "launches_counted": v.get("launches"),
                    "counters": {k: round(x, 1) for k, x in v.items() if k != "launches"},
                    "source": "measured in this run: rocprofv3 --pmc on 3 eager steps of the same workload (bench.pmc_counters_live)"}
    return {"hbm_bytes_per_launch": None, "source": "not measured: no launch of %s in the counter passes" % symbol}


def insitu_gemm_times(step, reps=6, rider_rows=None, rider_rows_launched=None):
    """Duration of every gemm_f16_kernel launch INSIDE the step, measured live with HIP events on the launch stream: the
    library launches each of them with a start / stop event pair (hipExtLaunchKernelGGL, cc_debug_gemm_timing_*: the events
    receive the dispatch's begin / end timestamps; the whole step is enqueued by one C call, far faster than the GPU drains
    it, so the launches run back to back between their real neighbours as in the captured graph).  -> {symbol: dict(us, launches_per_step, flops_per_step, shapes)}; flops count
    both problems of a paired launch (ViT carrier + text rider); the rider is counted with the rows it COMPUTES
    (`rider_rows`: the compacted captions, read from the device by its tiles) where the launch is sized for
    `rider_rows_launched` (captions x words)."""
    import ctypes
    from centerclip_amd import _lib as L
    lib = L.lib()
    lib.cc_debug_gemm_timing_begin.argtypes = [ctypes.c_int]
    lib.cc_debug_gemm_timing_read.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    assert lib.cc_debug_gemm_timing_begin(400 * reps) == 0
    try:
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
    finally:
        n = lib.cc_debug_gemm_timing_end()
    us, info = ctypes.c_float(), (ctypes.c_int * 12)()
    out = {}
    for i in range(n):
        assert lib.cc_debug_gemm_timing_read(i, ctypes.byref(us), info) == 0
        bm, bn, wm, wn, epi, bk, m0, n0, k0, m1, n1, k1 = list(info)
        sym = "gemm_f16_kernel<%d, %d, %d, %d, %d, %d>" % (bm, bn, wm, wn, epi, bk)
        e = out.setdefault(sym, dict(launches=0, flops=0.0, shapes={}))
        e["launches"] += 1
        m1c = rider_rows if (rider_rows is not None and m1 == rider_rows_launched) else m1
        e["flops"] += 2.0 * m0 * n0 * k0 + 2.0 * m1c * n1 * k1
        key = "%dx%dx%d%s" % (m0, n0, k0, " + rider %dx%dx%d%s" % (m1, n1, k1, " (%d rows computed)" % m1c if m1c != m1 else "") if m1 else "")
        e["shapes"].setdefault(key, []).append(us.value)
    lib.cc_debug_gemm_timing_begin(0)
    for e in out.values():
        # per shape the MEDIAN launch (one event pair that spans a pre-empted or re-clocked dispatch - seen once: 6.7 ms for a
        # 55 us launch - must not pose as the symbol's duration); the symbol's time = sum over its shapes of median x launches
        e["us"] = sum(statistics.median(v) * len(v) for v in e["shapes"].values())
        e["avg_us"] = e["us"] / e["launches"]
        e["tflops"] = e["flops"] / e["us"] / 1e6
        e["launches_per_step"] = e["launches"] / reps
        e["us_per_step"] = e["us"] / reps
        e["shapes"] = {k: dict(launches_per_step=len(v) / reps, avg_us=round(statistics.median(v), 2), max_us=round(max(v), 2))
                       for k, v in e["shapes"].items()}
    return out


def gemm_roofline(c, device, insitu=None):
    """Time every distinct GEMM of one step alone (HIP events on the launch stream around replays of a hipGraph of
    back-to-back launches), group them by kernel symbol and return the symbol with the largest share of the step as the
    dominant kernel.  Algorithmic flops = 2*M*N*K."""
    from centerclip_amd import ops
    W, B, T = c["width"], c["B"], c["T"]
    L0, L1 = 50, c["K"] + 1
    M0, M1 = B * T * L0, B * c["T_new"] * L1
    n0, n1 = c["cluster_block"] - 1, 13 - c["cluster_block"]
    # (name, M, N, K, epilogue id, calls per step)
    shapes = [("patch_embed", B * T * 49, W, 3 * 32 * 32, 3, 1),
              ("in_proj+attention" if L0 <= 256 else "in_proj", M0, 3 * W, W, 8 if L0 <= 256 else 5, n0), ("out_proj", M0, W, W, 7, n0),
              ("c_fc", M0, 4 * W, W, 6, n0), ("c_proj", M0, W, 4 * W, 7, n0),
              ("in_proj+attention@clustered" if L1 <= 256 else "in_proj@clustered", M1, 3 * W, W, 8 if L1 <= 256 else 5, n1),
              ("out_proj@clustered", M1, W, W, 7, n1 - 1),
              ("c_fc@clustered", M1, 4 * W, W, 6, n1 - 1), ("c_proj@clustered", M1, W, 4 * W, 7, n1 - 1)]
    # (block 12 runs out_proj / c_fc / c_proj on the B * T_new CLS rows only - gemm_rows_kernel, 0.06 GFLOP, not listed)
    rows = []
    for name, M, N, K, epi, calls in shapes:
        a = torch.randn(M, K, device=device).half()
        w = (torch.randn(N, K, device=device) * K ** -0.5).half()
        bias = torch.randn(N, device=device)
        if epi in (5, 6, 8):                     # LayerNorm-folded consumer epilogue, statistics in 12 slots
            hres = torch.randn(M, K, device=device)
            h16, _, _ = ops.row_stats(hres)
            stats = torch.randn(M, 12, 2, device=device).abs()
            wf, c1, c2 = ops.fold_layernorm_linear(w.float(), bias, torch.ones(K, device=device), torch.zeros(K, device=device))
            if epi == 8:                         # ... with the attention of the tile's frames behind it (L0 / L1 tokens per frame)
                Ltok = L1 if name.endswith("@clustered") else L0
                fn = (lambda h16=h16, wf=wf, c1=c1, c2=c2, stats=stats, nseq=M // Ltok, Ltok=Ltok:
                      ops.inproj_attention_f16(h16, wf, c1, c2, stats, 12, nseq, Ltok, K // 64))
            else:
                fn = (lambda h16=h16, wf=wf, c1=c1, c2=c2, stats=stats, g=(epi == 6):
                      ops.linear_ln_f16(h16, wf, c1, c2, stats, 12, gelu=g))
        elif epi == 7:                           # residual epilogue that also emits centred fp16 rows + partial sums
            hres = torch.zeros(M, N, device=device)
            h16b = torch.empty(M, N, device=device, dtype=torch.float16)
            stb = torch.empty(M * 32 * 2, device=device)
            _, st_in, sh_in = ops.row_stats(torch.randn(M, N, device=device))
            sh_out = torch.empty(M, device=device)
            fn = (lambda a=a, w=w, bias=bias, hres=hres, h16b=h16b, stb=stb, st_in=st_in, sh_in=sh_in, sh_out=sh_out:
                  ops.linear_resid_stats_f16(a, w, bias, hres, h16=h16b, stats=stb, shift_in=sh_in,
                                             stats_in=st_in.view(-1, 1, 2), shift_out=sh_out))
        else:                                    # the patch GEMM's shape with the plain fp32 epilogue
            out = torch.zeros(M, N, device=device, dtype=torch.float32)
            fn = lambda a=a, w=w, bias=bias, out=out: ops.linear_f16(a, w, bias, "f32", out=out)
        ms = graph_time_ms(fn)
        flops = 2.0 * M * N * K
        rows.append(dict(kernel=kernel_symbol(M, N, K, 4 if epi == 3 else epi), role=name, M=M, N=N, K=K, calls_per_step=calls,
                         avg_us=ms * 1e3, tflops=flops / ms / 1e9, step_share_us=ms * 1e3 * calls))
    # dominant kernel = the SYMBOL with the largest total time in the step (several shapes may share an instantiation)
    by_sym = {}
    for r in rows:
        s = by_sym.setdefault(r["kernel"], dict(us=0.0, flops=0.0, launches=0, roles=[]))
        s["us"] += r["step_share_us"]
        s["flops"] += 2.0 * r["M"] * r["N"] * r["K"] * r["calls_per_step"]
        s["launches"] += r["calls_per_step"]
        s["roles"].append(r["role"])
    standalone = {k: dict(roles=v["roles"], step_share_us=round(v["us"], 1), tflops=round(v["flops"] / v["us"] / 1e6, 1),
                          frac=round(v["flops"] / v["us"] / 1e6 / MFMA_F16_PEAK_TFLOPS, 4)) for k, v in by_sym.items()}
    if insitu:
        # headline: the dominant symbol's launches as they run INSIDE the step (what rocprofv3 --kernel-trace --stats of
        # this command averages too: profiles/*_bench_kernel_stats.*)
        sym, dom = max(insitu.items(), key=lambda kv: kv[1]["us"])
        tf, avg_us, n_l = dom["tflops"], dom["avg_us"], dom["launches_per_step"]
        flops_per_launch, share = dom["flops"] / dom["launches"], dom["us_per_step"]
        roles = standalone.get(sym, {}).get("roles", [])
        how = ("in situ: every launch of the symbol inside the eagerly enqueued step carries a start / stop HIP event "
               "(hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps); per shape the median of its launches over 6 steps")
    else:
        sym, dom = max(by_sym.items(), key=lambda kv: kv[1]["us"])
        tf, avg_us, n_l = dom["flops"] / dom["us"] / 1e6, dom["us"] / dom["launches"], dom["launches"]
        flops_per_launch, share, roles = dom["flops"] / dom["launches"], dom["us"], dom["roles"]
        how = "stand-alone: hipGraph of back-to-back launches of each shape (no text rider, no neighbours)"
    tr = pmc_traffic(sym)
    roof = dict(bound="mfma", kernel=sym, roles=roles, achieved=round(tf, 1), peak=MFMA_F16_PEAK_TFLOPS,
                unit="TFLOP/s", frac=round(tf / MFMA_F16_PEAK_TFLOPS, 4), measured=how,
                traffic=tr["hbm_bytes_per_launch"] if tr else None,
                traffic_unit=("bytes per launch across the L2 <-> fabric interface (Infinity-Cache hits included): 128 B x (TCC_EA0_RDREQ - "
                              "32B - 64B) + 64 B x RDREQ_64B + 32 B x RDREQ_32B + 64 B x WRREQ_64B + 32 B x (WRREQ - WRREQ_64B), "
                              "averaged over the symbol's launches inside the step; counter passes of this invocation"),
                traffic_detail=tr, avg_launch_us=round(avg_us, 2), launches_per_step=n_l,
                algorithmic_flops_per_launch=flops_per_launch,
                step_share_us=round(share, 1),
                by_symbol_in_situ={k: dict(avg_us=round(v["avg_us"], 2), launches_per_step=v["launches_per_step"],
                                           step_share_us=round(v["us_per_step"], 1), tflops=round(v["tflops"], 1),
                                           frac=round(v["tflops"] / MFMA_F16_PEAK_TFLOPS, 4), shapes=v["shapes"])
                                   for k, v in (insitu or {}).items()},
                by_symbol_stand_alone=standalone)
    gemm_us = sum(r["step_share_us"] for r in rows)
    gemm_flops = sum(2.0 * r["M"] * r["N"] * r["K"] * r["calls_per_step"] for r in rows)
    return roof, rows, gemm_us, gemm_flops


def forward_config_bench(key, device, seed=0):
    """One of the other BASELINE.json towers (cfg 3 / 4 / 5) at its per-GPU batch: the same step as the headline (both towers
    in one enqueue -> similarity logits), captured into a hipGraph and replayed; clips/s, whole-step fraction of the fp16
    MFMA peak on SURVEY 8(d)'s algorithmic flops, and the launches of every GEMM symbol inside the step (HIP event pairs,
    as `roofline` does for the headline).  The kernel tables of the same steps: profiles/r05_forward_<key>_kernel_stats.txt
    (`rocprofv3 --kernel-trace --stats -- python bench.py --workload <key> --no-extras --no-cpu-baseline`)."""
    from centerclip_amd.clip4clip import CLIP4Clip
    c = FORWARD_CFGS[key]
    sd = random_state_dict(c, seed=seed)
    model = CLIP4Clip.from_state_dict(dict(sd), task_config(c)).to(device).eval()
    ids, amask, video, vmask = synthetic_batch(c, device, seed=500 + seed)
    tt = torch.zeros_like(ids)

    def step():
        out = model(ids, tt, amask, video, vmask)
        return model.get_similarity_logits(out["sequence_output"], out["visual_output"], amask, vmask)[0]
    with torch.no_grad():
        logits = step()
        torch.cuda.synchronize()
        assert logits.shape == (c["B"], c["B"]) and bool(torch.isfinite(logits).all())
        ms = graph_time_ms(step, launches=1, replays=max(5, int(200 / max(1.0, 0.5 * c["B"]))))
        rows_text = int((ids.argmax(dim=-1) + 1).sum())
        ins = insitu_gemm_times(step, reps=3, rider_rows=rows_text, rider_rows_launched=int(ids.numel()))
    flops = algorithmic_flops_per_clip(c) * c["B"]
    sym, dom = max(ins.items(), key=lambda kv: kv[1]["us"])
    gemm_us = sum(v["us_per_step"] for v in ins.values())
    res = dict(workload=c["name"], ms_per_step=round(ms, 3), clips_per_s=round(c["B"] / ms * 1e3, 1), launch="hipGraph replay",
               algorithmic_gflop_per_clip=round(flops / c["B"] / 1e9, 1),
               whole_step_tflops=round(flops / ms / 1e9, 1), whole_step_frac_of_f16_mfma_peak=round(flops / ms / 1e9 / MFMA_F16_PEAK_TFLOPS, 4),
               gemm_launch_time_share_of_step=round(gemm_us / (ms * 1e3), 3),
               roofline=dict(bound="mfma", kernel=sym, achieved=round(dom["tflops"], 1), peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=round(dom["tflops"] / MFMA_F16_PEAK_TFLOPS, 4), avg_launch_us=round(dom["avg_us"], 2),
                             launches_per_step=dom["launches_per_step"], step_share_us=round(dom["us_per_step"], 1),
                             measured="in situ: HIP event pair around every launch of the symbol inside the eagerly enqueued step"),
               by_symbol_in_situ={k: dict(avg_us=round(v["avg_us"], 2), launches_per_step=v["launches_per_step"],
                                          step_share_us=round(v["us_per_step"], 1), tflops=round(v["tflops"], 1),
                                          frac=round(v["tflops"] / MFMA_F16_PEAK_TFLOPS, 4), shapes=v["shapes"]) for k, v in ins.items()})
    del model, video
    torch.cuda.empty_cache()
    return res


def cluster_bench(c, device, iters=30):
    """token-cluster Mtokens/s: the op alone on frame-major activations of one config's shape."""
    from centerclip_amd.cluster import TokenClusterInter
    B, T, Tn, K, n = c["B"], c["T"], c["T_new"], c["K"], c["n"]
    W = 768
    x = torch.randn(B * T, 1 + n, W, device=device)
    mod = TokenClusterInter(before_cluster_num=n, cluster_num=K, before_block_frames=T, after_block_frames=Tn,
                            original_frame=T, threshold=1e-6, iter_limit=100, split_size=c["split"], norm_p=2.0)
    ms = graph_time_ms(lambda: mod.cluster_frame_major(x, keep_ids=False), launches=10, replays=max(2, iters // 10))
    P, N = B * Tn, (T // Tn) * n
    tokens = P * N
    alg_bytes = P * N * W * 4 + P * K * W * 4 + P * K * 8
    return dict(mtokens_per_s=round(tokens / ms / 1e3, 2), us_per_call=round(ms * 1e3, 1), problems=P, tokens_per_problem=N,
                roofline=dict(bound="hbm", achieved=round(alg_bytes / ms / 1e6, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                              frac=round(alg_bytes / ms / 1e6 / HBM_PEAK_GBS, 4), algorithmic_bytes_per_launch=alg_bytes))


def spectral_cluster_bench(c, device):
    """The same op with cluster_algo='spectral' (heat-kernel graph -> Laplacian -> eigensolver, eig.hip -> k-medoids on the
    embedding -> gather), one config's shape; eager launches between two events (the decomposition dominates: ms)."""
    from centerclip_amd.cluster import TokenClusterInter
    B, T, Tn, K, n = c["B"], c["T"], c["T_new"], c["K"], c["n"]
    x = torch.randn(B * T, 1 + n, 768, device=device) * 0.05
    mod = TokenClusterInter(algorithm="spectral", before_cluster_num=n, cluster_num=K, before_block_frames=T,
                            after_block_frames=Tn, original_frame=T, threshold=1e-6, iter_limit=100, split_size=c["split"],
                            norm_p=2.0, spectral_sigma=2.0)
    ms = event_time_ms(lambda: mod.cluster_frame_major(x, keep_ids=False), 5)
    return dict(ms_per_call=round(ms, 2), mtokens_per_s=round(B * T * n / ms / 1e3, 2), problems=B * Tn,
                tokens_per_problem=(T // Tn) * n, launch="eager")


def cluster_pmc_traffic():
    """Fabric bytes of the token-cluster call inside the step = sum over its kernels, from this invocation's counter passes
    (None when they could not run)."""
    data = pmc_counters_live()
    if "error" in data:
        return None
    total, seen = 0.0, 0
    for key in ("gram_dist_kernel", "kmedoids_select_kernel"):     # K1, K2 (K0 is folded into K1, K3 into K2's tail)
        for name, v in data.items():
            if key in name:
                total += sum(pmc_bytes(v))
                seen += 1
                break
    return round(total) if seen == 2 else None


def similarity_bench(device, world=1):
    """pairwise-similarities/s: 10k texts x 1k videos (3 segments each): pooling / normalising into split fp16 planes + ONE
    fp16 MFMA GEMM over the K-concatenated planes (3 products per algorithmic multiply-add).  world > 1:
    rows sharded over ranks (dist.sharded_similarity with the HIP kernel), time = max over ranks."""
    from centerclip_amd import ops, dist as ccdist, torch_ops as T_
    Nt, Nv, Tn, E = 10000, 1000, 3, 512
    g = torch.Generator().manual_seed(11)
    t = torch.randn(Nt, E, generator=g).to(device)
    v = torch.randn(Nv, Tn, E, generator=g).to(device)
    m = torch.ones(Nv, Tn, dtype=torch.long, device=device)
    parts = None
    if world == 1:
        ms = event_time_ms(lambda: ops.loose_similarity(t, v, m, 1.0), 20)
        # the evaluation loop's form: operand planes written when the batches are encoded, the matrix = the GEMM alone
        tp = torch.ops.centerclip.normalize_rows_planes(t, False)
        vp = torch.zeros(T_.padded_video_rows(Nv), 3 * E, device=device, dtype=torch.float16)
        vp[:Nv] = torch.ops.centerclip.video_pool_normalize_planes(v, m)
        ms_gemm = graph_time_ms(lambda: torch.ops.centerclip.scaled_dot_planes(tp, vp, Nv, 2.718281828), launches=10, replays=3)
        ms_prep = graph_time_ms(lambda: (torch.ops.centerclip.normalize_rows_planes(t, False),
                                         torch.ops.centerclip.video_pool_normalize_planes(v, m)), launches=10, replays=3)
        parts = dict(gemm_from_cached_planes_us=round(ms_gemm * 1e3, 1), plane_writing_us=round(ms_prep * 1e3, 1),
                     note="eval_epoch writes the planes batch by batch with the encoders' outputs; its final matrix costs the GEMM",
                     pairs_per_s_gemm_alone=round(Nt * Nv / ms_gemm * 1e3, 0),
                     gemm_issued_f16_mfma_frac=round(3 * 2.0 * Nt * Nv * E / ms_gemm / 1e9 / MFMA_F16_PEAK_TFLOPS, 4))
        # fewer fp16 products per multiply-add (scaled_dot_planes(..., products)): time of the GEMM alone and the error of the
        # cosine matrix against float64 on the same unit rows (the contract asks 1e-3 of similarities)
        tn = (t.double() / t.double().norm(dim=-1, keepdim=True))
        vh = v.double() / v.double().norm(dim=-1, keepdim=True)
        vb = vh.mean(dim=1)
        exact = tn @ (vb / vb.norm(dim=-1, keepdim=True)).t()
        prods = {}
        for pr_ in (3, 2, 1):
            msp = graph_time_ms(lambda pr_=pr_: torch.ops.centerclip.scaled_dot_planes(tp, vp, Nv, 1.0, pr_), launches=10, replays=3)
            err = (torch.ops.centerclip.scaled_dot_planes(tp, vp, Nv, 1.0, pr_).double() - exact).abs()
            prods[str(pr_)] = dict(gemm_us=round(msp * 1e3, 1), pairs_per_s=round(Nt * Nv / msp * 1e3, 0),
                                   max_abs_err_vs_float64=float("%.3g" % float(err.max())), rms_err=float("%.3g" % float((err ** 2).mean().sqrt())))
        parts["products"] = prods
        parts["products_note"] = ("3 (default everywhere): hi.hi + hi.lo + lo.hi, both operands to 22 bits; 2: fp16(text) x video to 22 bits; "
                                  "1: fp16 x fp16 - eval_epoch(..., similarity_products=p)")
    else:
        t0, t1 = ccdist.shard_rows(Nt)
        v0, v1 = ccdist.shard_rows(Nv)
        tl = ops.normalize_rows(t[t0:t1])

        def run():
            pooled = ops.video_pool_normalize(v[v0:v1], m[v0:v1])          # this rank's videos
            return ccdist.sharded_similarity(tl, pooled, Nv, 2.718281828)   # all-gather of [Nv, E] + local NT GEMM
        ms = event_time_ms(run, 10)
        tt = torch.tensor([ms], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms = float(tt)
    flops = 2.0 * Nt * Nv * E
    # the NT GEMM runs 3 fp16 MFMA products per algorithmic multiply-add (hi.hi + hi.lo + lo.hi of 22-bit split operands)
    return dict(pairs_per_s=round(Nt * Nv / ms * 1e3, 0), us_per_call=round(ms * 1e3, 1),
                algorithmic_tflops=round(flops / ms / 1e9, 2), issued_f16_mfma_tflops=round(3 * flops / ms / 1e9, 2),
                frac_of_f16_mfma_peak=round(3 * flops / ms / 1e9 / MFMA_F16_PEAK_TFLOPS / world, 4),
                algorithmic_bytes=int((Nt + Nv * Tn) * E * 4 + Nt * Nv * 4),
                frac_of_hbm_peak=round(((Nt + Nv * Tn) * E * 4 + Nt * Nv * 4) / ms / 1e6 / HBM_PEAK_GBS / world, 4),
                sharding="rows over %d ranks, videos all-gathered (%.1f MB)" % (world, Nv * E * 4 / 1e6) if world > 1 else "single GPU",
                parts=parts)


def pcie_inclusive_bench(model, c, device, steps=150):
    """The same step fed FROM THE HOST: decoder-layout uint8 frames (N3) + ids / masks in pinned memory, staged by
    centerclip_amd.feeder.DeviceFeeder (two device slots filled on a copy stream while the encoders run on the other slot, one
    captured hipGraph per slot).  -> clips/s with the copies overlapped, and with copy and compute serialised."""
    from centerclip_amd.feeder import DeviceFeeder
    g = torch.Generator().manual_seed(5)
    host = []
    for i in range(3):
        ids, amask, _, vmask = [t.cpu() for t in synthetic_batch(c, "cpu", seed=300 + i)]
        u8 = torch.randint(0, 256, (c["B"], 1, c["T"], c["res"], c["res"], 3), dtype=torch.uint8, generator=g)
        host.append(tuple(t.pin_memory() for t in (ids, torch.zeros_like(ids), amask, u8, vmask)))
    bytes_per_step = sum(t.numel() * t.element_size() for t in host[0])
    feeder = DeviceFeeder(device, depth=2)

    def step(bufs):
        ids, seg, amask, video, vmask = bufs
        out = model(ids, seg, amask, video, vmask)
        return model.get_similarity_logits(out["sequence_output"], out["visual_output"], amask, vmask)[0]

    graphs = {}
    stream_batches = (host[i % len(host)] for i in range(steps + 4))
    torch.cuda.synchronize()
    t0 = None
    done = 0
    for k, bufs in feeder(stream_batches):
        if k not in graphs:                              # first visit of a slot: warm up + capture on its (stable) tensors
            step(bufs)
            torch.cuda.synchronize()
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                graphs[k] = (gph, step(bufs))
            torch.cuda.synchronize()
            continue
        if t0 is None:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        graphs[k][0].replay()
        done += 1
    torch.cuda.synchronize()
    overlapped = (time.perf_counter() - t0) / done
    # serialised reference: copy, wait, compute, wait
    dev_bufs = tuple(h.to(device) for h in host[0])
    step(dev_bufs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n = 20
    for i in range(n):
        for d, h in zip(dev_bufs, host[i % len(host)]):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        step(dev_bufs)
        torch.cuda.synchronize()
    serial = (time.perf_counter() - t1) / n
    return dict(clips_per_s=round(c["B"] / overlapped, 1), ms_per_step=round(overlapped * 1e3, 3), steps=done,
                input="uint8 HWC frames + ids / masks in pinned host memory, %.1f MB per step" % (bytes_per_step / 1e6),
                staging="centerclip_amd.feeder.DeviceFeeder: 2 device slots, H2D on a copy stream under the previous step, one hipGraph per slot",
                h2d_gb_per_s_needed=round(bytes_per_step / overlapped / 1e9, 1),
                serialised_copy_then_compute={"ms_per_step": round(serial * 1e3, 3), "clips_per_s": round(c["B"] / serial, 1),
                                              "launch": "eager"})


def cpu_baseline(c, state_dict):
    """The reference path restated in plain PyTorch on the host CPU (oracle/, kind 'port'): text tower +
    ViT with the literal k-medoids + meanP similarity, all cores, on a bounded sample of the same workload."""
    from oracle import clip_oracle as clo
    cores = os.cpu_count() or 1
    torch.set_num_threads(min(cores, 64))
    used = torch.get_num_threads()
    plan = {c["cluster_block"] - 1: (c["T_new"], c["K"])}

    def run(nclips, seed):
        g = torch.Generator().manual_seed(seed)
        video = torch.randn(nclips * c["T"], 3, c["res"], c["res"], generator=g)
        ids = torch.randint(1, 49405, (nclips, c["words"]), generator=g)
        ids[:, 0], ids[:, -1] = 49406, 49407
        t0 = time.perf_counter()
        with torch.no_grad():
            v = clo.visual_forward(state_dict, video, c["T"], cluster_plan=plan).view(nclips, c["T_new"], -1)
            t = clo.text_forward(state_dict, ids).view(nclips, 1, -1)
            clo.loose_similarity(t, v, torch.ones(nclips, c["T_new"], dtype=torch.long), float(state_dict["logit_scale"]))
        return time.perf_counter() - t0

    run(1, 0)                                            # warm-up (thread pools, allocator)
    t1 = run(1, 1)
    n = int(max(1, min(c["B"], round(12.0 / max(t1, 1e-3)))))
    t = run(n, 2)
    batches = 1
    while t < 10.0 and batches < 8:                      # a bounded sample of ~10-30 s: whole batches of the workload
        t += run(n, 2 + batches)
        batches += 1
    return dict(value=round(n * batches / t, 3), unit="clips/s", cores=used, kind="port",
                sample="%d batch(es) of %d clip(s) x 12 frames + %d caption(s) through oracle/clip_oracle.py (fp32, literal "
                       "k-medoids), %.1f s of CPU work; single clip %.2f s" % (batches, n, n, t, t1))


def cpu_baseline_cluster(c):
    from oracle import cluster_oracle as co
    X = torch.randn(c["B"] * c["T_new"], (c["T"] // c["T_new"]) * 49, c["width"])
    co.literal_batch_kmedoids_with_split(X[:16], c["K"], "euclidean", 1e-6, 100, True, 2.0, 16, False)
    t0 = time.perf_counter()
    co.literal_batch_kmedoids_with_split(X, c["K"], "euclidean", 1e-6, 100, True, 2.0, 16, False)
    t = time.perf_counter() - t0
    return dict(value=round(X.shape[0] * X.shape[1] / t / 1e6, 4), unit="Mtokens/s", cores=torch.get_num_threads(), kind="port",
                sample="one call on [48,196,768] fp32, %.2f s" % t)


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start N ranks on this node."""
    share = os.environ.get("CC_BENCH_SHARE_GPU") == "1"
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < a.gpus and not share:
        sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible - refusing to report a %d-GPU number from fewer "
                         "devices\n" % (a.gpus, ndev, a.gpus))
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def two_in_flight_bench(c, sd, device, pairs=150):
    """Throughput with TWO independent B = 16 batches in flight: a second model instance (its own workspace, the same weights), one
    hipGraph per instance, replayed on two streams.  A step's low-occupancy phases (the k-medoids selection on 48 of the 256
    CUs, the last block's few-rows launches, heads, launch tails) then run under the other batch's GEMMs.  Reported beside the
    headline, which stays one batch in flight: per-kernel durations - what `roofline` and the rocprofv3 profile are about -
    are not meaningful while two steps share the chip."""
    keep, graphs, streams = two_in_flight_graphs(c, sd, device)

    def timed(fn, n):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def one():
        with torch.cuda.stream(streams[0]):
            graphs[0].replay()

    def two():
        for s_ in range(2):
            with torch.cuda.stream(streams[s_]):
                graphs[s_].replay()
    ms1 = min(timed(one, 2 * pairs) for _ in range(2))
    ms2 = min(timed(two, pairs) for _ in range(2)) / 2
    assert bool(torch.isfinite(keep[0][-1]).all()) and bool(torch.isfinite(keep[1][-1]).all())
    return {"ms_per_step": round(ms2, 3), "clips_per_s": round(c["B"] / ms2 * 1e3, 1),
            "one_in_flight_same_harness_ms_per_step": round(ms1, 3),
            "how": "two model instances (same weights, own workspaces), one hipGraph each, replayed alternately on two streams"}


def two_in_flight_graphs(c, sd, device):
    """-> (keep, graphs, streams): two model instances, each warmed up and captured on a stream of its own."""
    from centerclip_amd.clip4clip import CLIP4Clip
    keep, graphs, streams = [], [], [torch.cuda.Stream(device), torch.cuda.Stream(device)]
    for s_ in range(2):
        m = CLIP4Clip.from_state_dict(dict(sd), task_config(c)).to(device).eval()
        ids, amask, video, vmask = synthetic_batch(c, device, seed=700 + s_)
        tt = torch.zeros_like(ids)

        def step(m=m, ids=ids, tt=tt, amask=amask, video=video, vmask=vmask):
            out = m(ids, tt, amask, video, vmask)
            return m.get_similarity_logits(out["sequence_output"], out["visual_output"], amask, vmask)[0]
        with torch.no_grad(), torch.cuda.stream(streams[s_]):
            for _ in range(3):
                step()
            streams[s_].synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[s_]):
                out = step()
        torch.cuda.synchronize()
        keep.append((m, ids, amask, video, vmask, tt, step, out))   # (a captured graph holds raw addresses of all of these)
        graphs.append(g)
    return keep, graphs, streams


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--min-seconds", type=float, default=2.0, help="keep timing K-step windows until this much timed work ran")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of a captured hipGraph replay")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / cluster / similarity side measurements")
    ap.add_argument("--in-flight", type=int, default=1, choices=[1, 2],
                    help="2: the timed steps alternate between two model instances / hipGraphs on two streams (two batches in "
                         "flight, the serving-loop form; 1 GPU only).  The default, and the judged line, is 1: `roofline` and the "
                         "rocprofv3 profile are per-kernel statements, which two overlapping steps blur")
    ap.add_argument("--workload", default="cfg2", choices=sorted(FORWARD_CFGS),
                    help="the configuration the timed steps run (default cfg2 = BASELINE.json's metric configuration, the judged "
                         "line); cfg3 / cfg4 / cfg5 time the other towers at their per-GPU batch - used for their rocprofv3 tables")
    a = ap.parse_args()
    if a.workload != "cfg2":             # a profiling aid: the step alone (the side measurements are cfg2's)
        a.no_extras = a.no_cpu_baseline = True

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d started with WORLD_SIZE=%d: launch it as `python bench.py --gpus N` or under "
                         "torch.distributed.run with --nproc-per-node N" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    share = os.environ.get("CC_BENCH_SHARE_GPU") == "1"   # dev aid: N ranks on ONE GPU over gloo, to exercise the N>1 code path
    if share:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from centerclip_amd.clip4clip import CLIP4Clip
    from centerclip_amd import dist as ccdist
    c = FORWARD_CFGS[a.workload]
    sd = random_state_dict(c, seed=0)                    # same weights on every rank
    model = CLIP4Clip.from_state_dict(dict(sd), task_config(c)).to(device).eval()
    ids, amask, video, vmask = synthetic_batch(c, device, seed=100 + rank)
    token_type = torch.zeros_like(ids)                   # an input of the reference signature (unused by the path)
    scale = float(sd["logit_scale"])
    sink = ccdist.PackedFeatures(c["B"], c["T_new"], 512, device, world) if world > 1 else None

    def step1():                                         # single GPU: the reference's call sequence (main.py:444,518)
        out = model(ids, token_type, amask, video, vmask)
        logits, *_ = model.get_similarity_logits(out["sequence_output"], out["visual_output"], amask, vmask)
        return logits

    def towers_n():                                      # N GPUs: both towers write into the preallocated record
        model.encode_into(sink, ids, video, vmask)

    def tail_n():                                        # exchange step (ONE all-gather) + this rank's row block
        sink.gather()
        return sink.logits(sink.seq, scale)

    graph = None
    with torch.no_grad():
        for _ in range(max(a.warmup, 1)):
            if world == 1:
                logits = step1()
            else:
                towers_n()
                logits = tail_n()
        torch.cuda.synchronize()
        if not a.no_graph:
            # capture into a hipGraph: removes ~70 host launches per step from the critical path.  Inputs stay
            # resident, so a replay IS one pass of the hot path over the batch.  1 GPU: the whole step; N GPUs: both
            # towers (the RCCL all-gather and the similarity launch that follows it are issued eagerly after it).
            try:
                gph = torch.cuda.CUDAGraph()
                # thread_local: the RCCL watchdog thread of an initialised process group may query events meanwhile
                with torch.cuda.graph(gph, capture_error_mode="thread_local" if world > 1 else "global"):
                    captured = step1() if world == 1 else towers_n()
                gph.replay()
                torch.cuda.synchronize()
                graph = gph
            except Exception as exc:          # noqa: BLE001 - report and fall back to eager launches
                sys.stderr.write("graph capture failed (%s); timing eager launches\n" % exc)
                graph = None
                torch.cuda.synchronize()
        flight = None
        if a.in_flight == 2:
            if world != 1 or graph is None:
                raise SystemExit("--in-flight 2: one GPU, graph launch")
            flight = two_in_flight_graphs(c, sd, device)          # (keeps the models / inputs / outputs of both graphs alive)
            turn = [0]

            def run_two():
                i = turn[0] & 1
                turn[0] += 1
                with torch.cuda.stream(flight[2][i]):
                    flight[1][i].replay()
                return flight[0][i][-1]
        if flight is not None:
            run = run_two
        elif world == 1:
            run = step1 if graph is None else (lambda: (graph.replay(), captured)[1])
        elif graph is None:
            run = lambda: (towers_n(), tail_n())[1]
        else:
            run = lambda: (graph.replay(), tail_n())[1]
        for _ in range(a.warmup):
            run()
        windows = []
        total = 0.0
        while True:
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                out_ = run()
                if out_ is not None:
                    logits = out_
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if world > 1:                    # max over ranks (also keeps every rank on the same number of windows)
                tt = torch.tensor([el], device=device, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt)
            windows.append(el)
            total += el
            if total >= a.min_seconds or len(windows) >= 400:
                break
    assert logits.shape == (c["B"], c["B"] * world) and bool(torch.isfinite(logits).all())
    elapsed = statistics.median(windows)

    extras = {}
    if world > 1:                        # always reported at N > 1: how many ranks RCCL carried and what the exchange step cost
        with torch.no_grad():
            ms = event_time_ms(sink.gather, 50)
        tt = torch.tensor([ms], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        extras["feature_all_gather"] = dict(bytes_per_rank=sink.rec, bytes_gathered=sink.bytes_per_gather,
                                            us_per_call=round(float(tt) * 1e3, 1), backend=dist.get_backend(),
                                            ranks=dist.get_world_size(), collectives_per_step=1)
    if not a.no_extras:
        with torch.no_grad():
            tc = {name: cluster_bench(s, device, iters=30 if name == "cfg2" else 10) for name, s in CLUSTER_SHAPES.items()}
            if world > 1:                    # replicated op: the node's rate is the sum of the ranks' rates
                for name in tc:
                    tt = torch.tensor([tc[name]["mtokens_per_s"]], device=device, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.SUM)
                    tc[name]["mtokens_per_s_all_ranks"] = round(float(tt), 2)
            extras["token_cluster"] = tc
            if world == 1:
                extras["token_cluster_spectral"] = {name: spectral_cluster_bench(sh, device) for name, sh in CLUSTER_SHAPES.items()}
            extras["similarity_10k_x_1k"] = similarity_bench(device, world)
            if world == 1 and a.workload == "cfg2":      # the other BASELINE.json towers, one GPU's share each
                extras["forward_other_configs"] = {k: forward_config_bench(k, device) for k in ("cfg3", "cfg4", "cfg5")}
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        clips = c["B"] * world * a.steps
        res = {"metric": "clips/sec (ViT-B/32, 12f)", "value": round(clips / elapsed, 2), "unit": "clips/s",
               "n_gpus": world, "collective_backend": (dist.get_backend() if world > 1 else None),
               "ranks_in_communicator": (dist.get_world_size() if world > 1 else 1),
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16 MFMA operands, fp32 accumulate/residual/LN/softmax; cluster + similarity fp32",
               "data": "synthetic (N(0,1) frames, random token ids, random-init weights with CLIP init statistics rounded through fp16)",
               "timing": {"windows": len(windows), "steps_per_window": a.steps, "statistic": "median window (max over ranks per window)",
                          "first_window_ms_per_step": round(windows[0] / a.steps * 1e3, 3),
                          "min_window_ms_per_step": round(min(windows) / a.steps * 1e3, 3),
                          "max_window_ms_per_step": round(max(windows) / a.steps * 1e3, 3),
                          "timed_seconds": round(total, 3)},
               "launch": ("two hipGraphs replayed alternately on two streams (two batches in flight, --in-flight 2)" if a.in_flight == 2 else
                          ("hipGraph replay" if world == 1 else "hipGraph replay (towers) + eager all-gather / similarity") if graph is not None else "eager launches"),
               "config": {"workload": c["name"], "global_batch": c["B"] * world,
                          "parallelism": "dp%d (clips sharded, one RCCL all-gather of preallocated feature records)" % world if world > 1 else "single GPU"}}
        res.update(extras)
        if "token_cluster" in res:
            res["token_cluster"]["cfg2"]["roofline"]["traffic"] = cluster_pmc_traffic() if (world == 1 and not a.no_extras) else None
            res["token_cluster"]["cfg2"]["roofline"]["traffic_unit"] = ("bytes per call across the L2 <-> fabric interface, sum over K1 + K2 "
                                                                         "inside the step (TCC_EA0 request counters, counter passes of this invocation)")
            res["token_cluster_mtokens_per_s"] = res["token_cluster"]["cfg2"].get("mtokens_per_s_all_ranks",
                                                                                  res["token_cluster"]["cfg2"]["mtokens_per_s"])
        if world == 1 and not a.no_extras:
            with torch.no_grad():
                rows_text = int((ids.argmax(dim=-1) + 1).sum())
                roof, rows, gemm_us, gemm_flops = gemm_roofline(c, device, insitu_gemm_times(step1, rider_rows=rows_text,
                                                                                              rider_rows_launched=int(ids.numel())))
                res["roofline"] = roof
                res["gemm_breakdown"] = [{k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]
                res["gemm_time_share_of_step"] = round(gemm_us / (ms_per_step * 1e3), 3)
                res["forward_algorithmic_tflops"] = round(gemm_flops * 1.0 / (ms_per_step * 1e-3) / 1e12, 1)
                # the same step with the text tower on all 16 x 32 rows (caption compaction off), for transparency
                with model.clip.row_policy(all_text_rows=True):
                    ms_dense = graph_time_ms(step1, launches=1, replays=20)
                lens = (ids.argmax(dim=-1) + 1).float()
                res["text_rows"] = {"policy": "captions compacted to their EOT on the device: tokens behind the EOT cannot reach the "
                                              "caption's feature (causal mask, EOT row projected) - features bit-identical to all rows",
                                    "rows_computed": int(lens.sum()), "rows_all": int(ids.numel()),
                                    "ms_per_step_all_rows": round(ms_dense, 3)}
                with model.clip.row_policy(all_last_block_rows=True):
                    ms_all12 = graph_time_ms(step1, launches=1, replays=20)
                res["last_block_rows"] = {"policy": "the last block of each tower computes out_proj / c_fc / c_proj for the rows its "
                                                    "projection head reads (CLS of every frame, EOT of every caption); features "
                                                    "agree with the all-rows form to the rounding of the fp16 intermediates (<= 2e-4 relative, tested)",
                                          "rows_computed": c["B"] * c["T_new"] + c["B"],
                                          "rows_all": c["B"] * c["T_new"] * (c["K"] + 1) + int(lens.sum()),
                                          "ms_per_step_all_rows": round(ms_all12, 3)}
                # the same step at other evaluation batch sizes (the headline stays B = 16, BASELINE cfg 2): eval batches are
                # independent, so batch_size_val is the user's to choose - the clustered blocks' 2,400-row GEMMs become 4,800 / 1,200
                other = {}
                for b2 in (8, 32):
                    c2 = dict(c, B=b2)
                    i2, m2, v2, vm2 = synthetic_batch(c2, device, seed=300 + b2)
                    z2 = torch.zeros_like(i2)

                    def step_b(i2=i2, m2=m2, v2=v2, vm2=vm2, z2=z2):
                        o2 = model(i2, z2, m2, v2, vm2)
                        return model.get_similarity_logits(o2["sequence_output"], o2["visual_output"], m2, vm2)[0]
                    ms_b = graph_time_ms(step_b, launches=1, replays=20)
                    other["B=%d" % b2] = {"ms_per_step": round(ms_b, 3), "clips_per_s": round(b2 / ms_b * 1e3, 1)}
                    del i2, m2, v2, vm2, z2
                model(ids, token_type, amask, video, vmask)           # back to the headline batch's workspace
                res["other_batch_sizes"] = other
                res["two_batches_in_flight"] = two_in_flight_bench(c, sd, device)
                # N3: the same step fed with decoder-layout uint8 frames (normalisation fused into the patch gather)
                u8 = torch.randint(0, 256, (c["B"], 1, c["T"], 224, 224, 3), dtype=torch.uint8, device=device)
                ms_u8 = event_time_ms(lambda: model(ids, torch.zeros_like(ids), amask, u8, vmask), 10)
                ms_f32 = event_time_ms(lambda: model(ids, torch.zeros_like(ids), amask, video, vmask), 10)
                res["pcie_inclusive"] = pcie_inclusive_bench(model, c, device)
                res["uint8_input"] = {"ms_per_forward_uint8_hwc": round(ms_u8, 3), "ms_per_forward_f32": round(ms_f32, 3),
                                      "input_bytes_per_clip": {"uint8": c["T"] * 3 * 224 * 224, "f32": c["T"] * 3 * 224 * 224 * 4},
                                      "launch": "eager"}
        else:
            res["roofline"] = None
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(c, sd)
            res["token_cluster_cpu_baseline"] = cpu_baseline_cluster(c)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
