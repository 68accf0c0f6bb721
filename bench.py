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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from bench_common import (MFMA_F16_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS, HBM_PEAK_GBS, CFG2, FORWARD_CFGS, CLUSTER_SHAPES,  # noqa: E402,F401
                          task_config, algorithmic_flops_per_clip, random_state_dict, synthetic_batch, event_time_ms, graph_time_ms)


DTYPE = "fp16 (MFMA operands; fp32 accumulate, residual stream, LN, softmax; cluster + similarity fp32)"
LINE_LIMIT = 4096                   # the driver keeps a tail of stdout: the judged line must fit into it whole


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def compact_line(res, detail_path=None):
    """The ONE stdout line of the contract from the full result dict: headline + roofline summary + cpu_baseline + the two other
    rates the metric names, nothing per-shape.  Pure (tests/test_bench_helpers.py builds it from a canned result and checks
    its size and that it round-trips through json)."""
    line = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data", "launch", "collective_backend", "ranks_in_communicator"))
    line["config"] = _pick(res.get("config"), ("workload", "global_batch", "parallelism"))
    roof = res.get("roofline")
    if roof:
        r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches_per_step",
                         "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch", "step_share_us"))
        r["traffic_source"] = (roof.get("traffic_detail") or {}).get("source")
        if roof.get("by_shape"):
            r["by_shape"] = [_pick(b, ("shape", "avg_us", "mfma_frac", "hbm_frac", "bound")) for b in roof["by_shape"]]
        line["roofline"] = r
    else:
        line["roofline"] = None
    cb = res.get("cpu_baseline")
    line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample")) if cb else None
    tc = (res.get("token_cluster") or {}).get("cfg2")
    if tc:
        line["token_cluster_mtokens_per_s"] = res.get("token_cluster_mtokens_per_s", tc.get("mtokens_per_s"))
        line["token_cluster"] = dict(us_per_call=tc.get("us_per_call"),
                                     roofline=_pick(tc.get("roofline"), ("bound", "achieved", "peak", "unit", "frac", "traffic",
                                                                         "algorithmic_bytes_per_launch")))
        tcb = res.get("token_cluster_cpu_baseline")
        if tcb:
            line["token_cluster"]["cpu_baseline"] = _pick(tcb, ("value", "unit", "cores", "kind"))
    sim = res.get("similarity_10k_x_1k")
    if sim:
        line["pairs_per_s"] = sim.get("pairs_per_s")
        line["similarity_10k_x_1k"] = _pick(sim, ("us_per_call", "form", "raw_features_us", "frac_of_hbm_peak", "frac_of_f16_mfma_peak", "sharding"))
    if "timing" in res:
        line["timing"] = _pick(res["timing"], ("windows", "statistic", "min_window_ms_per_step", "max_window_ms_per_step"))
    if "feature_all_gather" in res:
        line["feature_all_gather"] = _pick(res["feature_all_gather"], ("bytes_gathered", "us_per_call", "backend", "ranks",
                                                                       "collectives_per_step"))
    oc = res.get("forward_other_configs")
    if oc:
        line["other_configs"] = {k: dict(clips_per_s=v.get("clips_per_s"), ms_per_step=v.get("ms_per_step"),
                                         frac=v.get("whole_step_frac_of_f16_mfma_peak")) for k, v in oc.items()}
    tf = res.get("two_batches_in_flight")
    if tf:                            # the serving-loop form (eval_epoch(in_flight=2)): reported beside the headline, never as `value`
        line["two_batches_in_flight"] = _pick(tf, ("clips_per_s", "ms_per_step"))
    el = res.get("eval_loop")
    if el:                            # the product's eval_epoch from pinned host batches (copies + host launches included)
        line["eval_loop_clips_per_s"] = {f: {k: v["clips_per_s"] for k, v in r.items() if isinstance(v, dict)} for f, r in el.items()
                                         if isinstance(r, dict)}
    if "forward_algorithmic_tflops" in res:
        line["whole_step_tflops"] = res["forward_algorithmic_tflops"]
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    return line


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--min-seconds", type=float, default=2.0, help="keep timing K-step windows until this much timed work ran")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of a captured hipGraph replay")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / cluster / similarity side measurements")
    ap.add_argument("--pmc", action="store_true",
                    help="collect the L2 <-> fabric request counters of the step's kernels in THIS invocation (two rocprofv3 --pmc child "
                         "runs of 3 eager steps each, bench_side.pmc_counters_live) and save them next to the detail file; without it "
                         "`roofline.traffic` is read from the committed copy of such a run (profiles/, source named in the line)")
    ap.add_argument("--detail-json", default=None,
                    help="where the DETAIL goes (every side measurement, per-shape tables, raw counters; default "
                         "gpurun_out/bench_detail_n<N>.json).  stdout carries only the compact contract line")
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
        import bench_side as side
        side.PMC_LIVE = bool(a.pmc)
        with torch.no_grad():
            tc = {name: side.cluster_bench(s, device, iters=30 if name == "cfg2" else 10) for name, s in CLUSTER_SHAPES.items()}
            if world > 1:                    # replicated op: the node's rate is the sum of the ranks' rates
                for name in tc:
                    tt = torch.tensor([tc[name]["mtokens_per_s"]], device=device, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.SUM)
                    tc[name]["mtokens_per_s_all_ranks"] = round(float(tt), 2)
            extras["token_cluster"] = tc
            if world == 1:
                extras["token_cluster_spectral"] = {name: side.spectral_cluster_bench(sh, device) for name, sh in CLUSTER_SHAPES.items()}
            extras["similarity_10k_x_1k"] = side.similarity_bench(device, world)
            if world == 1 and a.workload == "cfg2":      # the other BASELINE.json towers, one GPU's share each
                extras["forward_other_configs"] = {k: side.forward_config_bench(k, device) for k in ("cfg3", "cfg4", "cfg5")}
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        clips = c["B"] * world * a.steps
        res = {"metric": "clips/sec (ViT-B/32, 12f)", "value": round(clips / elapsed, 2), "unit": "clips/s",
               "n_gpus": world, "collective_backend": (dist.get_backend() if world > 1 else None),
               "ranks_in_communicator": (dist.get_world_size() if world > 1 else 1),
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE,
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
            roofc = res["token_cluster"]["cfg2"]["roofline"]
            roofc["traffic"], roofc["traffic_source"] = side.cluster_pmc_traffic() if world == 1 else (None, "N > 1: not collected")
            roofc["traffic_unit"] = ("bytes per call across the L2 <-> fabric interface, sum over K1 + K2 inside the step "
                                     "(TCC_EA0 request counters)")
            res["token_cluster_mtokens_per_s"] = res["token_cluster"]["cfg2"].get("mtokens_per_s_all_ranks",
                                                                                  res["token_cluster"]["cfg2"]["mtokens_per_s"])
        if world == 1 and not a.no_extras:
            with torch.no_grad():
                rows_text = int((ids.argmax(dim=-1) + 1).sum())
                roof, rows, gemm_us, gemm_flops = side.gemm_roofline(c, device, side.insitu_gemm_times(step1, rider_rows=rows_text,
                                                                                                        rider_rows_launched=int(ids.numel())))
                res["roofline"] = roof
                res["gemm_breakdown"] = [{k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]
                res["gemm_time_share_of_step"] = round(gemm_us / (ms_per_step * 1e3), 3)
                res["forward_algorithmic_tflops"] = round(gemm_flops * 1.0 / (ms_per_step * 1e-3) / 1e12, 1)
                res.update(side.step_policies_and_variants(model, c, sd, device, step1, (ids, token_type, amask, video, vmask)))
        else:
            res["roofline"] = None
        if world == 1 and not a.no_cpu_baseline:
            import bench_side as side
            res["cpu_baseline"] = side.cpu_baseline(c, sd)
            res["token_cluster_cpu_baseline"] = side.cpu_baseline_cluster(c)
        else:
            res["cpu_baseline"] = None
        # the DETAIL (every side measurement, per-shape tables, raw counters) goes to a file; stdout carries ONE compact line
        detail_path = a.detail_json or os.path.join(ROOT, "gpurun_out", "bench_detail_n%d.json" % world)
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(res, f)
        except OSError as exc:
            sys.stderr.write("bench detail not written (%s)\n" % exc)
            detail_path = None
        line = compact_line(res, detail_path)
        s = json.dumps(line, separators=(",", ":"))
        assert len(s) < LINE_LIMIT, len(s)
        sys.stdout.flush()
        print(s, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
