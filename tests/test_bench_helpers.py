"""bench.py helpers (CPU only): the compact stdout line (round 6: the driver keeps a tail of stdout, so the judged line must be
small - the detail goes to a file), the byte arithmetic of the counter passes that fill `roofline.traffic`
(bench_side.pmc_counters: this invocation's with --pmc, else the committed copy), the name matching of kernel symbols against
rocprofv3's kernel names, and the dispatcher's tile -> symbol mapping."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name + "_module", os.path.join(ROOT, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _bench():
    return _load("bench_side")


def test_stdout_line_is_compact_and_complete():
    """The full result of a real run (round 5's 20 KB dict, the one the driver could not parse) -> the stdout line: under 4 KB,
    round-trips through json, carries every key of the contract with the roofline / cpu_baseline objects, and the numbers are
    the detail's numbers."""
    b = _load("bench")
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))
    assert len(json.dumps(full)) > 15000
    line = b.compact_line(full, os.path.join(ROOT, "gpurun_out", "bench_detail_n1.json"))
    s = json.dumps(line, separators=(",", ":"))
    assert len(s) < b.LINE_LIMIT <= 4096 and "\n" not in s
    d = json.loads(s)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "launch", "token_cluster_mtokens_per_s",
                "pairs_per_s"):
        assert key in d, key
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and d["config"]["workload"] == full["config"]["workload"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches_per_step"):
        assert d["roofline"][key] == full["roofline"][key], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert d["cpu_baseline"][key] == full["cpu_baseline"][key]
    assert d["token_cluster"]["roofline"]["bound"] == "hbm" and d["detail"] == "gpurun_out/bench_detail_n1.json"
    # a run without side measurements / on N ranks still yields a valid line
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config")}
    bare.update(roofline=None, cpu_baseline=None, feature_all_gather=dict(bytes_gathered=1, us_per_call=2.0, backend="nccl", ranks=8,
                                                                           collectives_per_step=1, bytes_per_rank=3))
    d2 = json.loads(json.dumps(b.compact_line(bare)))
    assert d2["roofline"] is None and d2["cpu_baseline"] is None and d2["feature_all_gather"]["ranks"] == 8 and d2["detail"] is None


def test_pmc_traffic_from_live_counters():
    """The calibration launches of profiles/r05_traffic_reconcile.txt as counter records: a 512 MiB copy (4,194,510 read requests
    of 128 B, 8,388,608 write requests of 64 B) and a GEMM symbol; a failed counter pass yields null + the reason, never a
    number from somewhere else; the record's source travels with the number."""
    b = _bench()
    b._PMC_CACHE["data"] = {
        "__amd_rocclr_copyBuffer": {"TCC_EA0_RDREQ_sum": 4194510.0, "TCC_EA0_RDREQ_32B_sum": 0.0, "TCC_EA0_RDREQ_64B_sum": 0.0,
                                    "TCC_EA0_RDREQ_128B_sum": 0.0, "TCC_EA0_WRREQ_sum": 8388608.0, "TCC_EA0_WRREQ_64B_sum": 8388608.0, "launches": 4},
        "void gemm_f16_kernel<256, 128, 4, 2, 7, 64>(GemmPair)": {"TCC_EA0_RDREQ_sum": 1010986.5, "TCC_EA0_RDREQ_32B_sum": 0.0,
                                                                      "TCC_EA0_RDREQ_64B_sum": 0.0, "TCC_EA0_WRREQ_sum": 719256.0,
                                                                      "TCC_EA0_WRREQ_64B_sum": 700184.0, "launches": 8},
        "void gram_dist_kernel<0>(float const*)": {"TCC_EA0_RDREQ_sum": 230000.0, "TCC_EA0_WRREQ_sum": 115000.0, "TCC_EA0_WRREQ_64B_sum": 115000.0, "launches": 3},
        "void kmedoids_select_kernel<true, 4>(float const*)": {"TCC_EA0_RDREQ_sum": 110000.0, "TCC_EA0_WRREQ_sum": 120000.0, "TCC_EA0_WRREQ_64B_sum": 118000.0, "launches": 3},
    }
    fetch, write = b.pmc_bytes(b._PMC_CACHE["data"]["__amd_rocclr_copyBuffer"])
    assert abs(fetch - 512 * 2 ** 20) < 1e5 and write == 512 * 2 ** 20
    tr = b.pmc_traffic("gemm_f16_kernel<256, 128, 4, 2, 7, 64>")
    assert tr["fetch_bytes"] == round(1010986.5 * 128) and tr["write_bytes"] == 700184 * 64 + (719256 - 700184) * 32
    assert tr["hbm_bytes_per_launch"] == tr["fetch_bytes"] + tr["write_bytes"] and tr["source"] == "test record"
    assert b.pmc_traffic("gemm_f16_kernel<1, 2, 3, 4, 5, 6>")["hbm_bytes_per_launch"] is None
    total, src = b.cluster_pmc_traffic()
    assert src == "test record" and total == round(230000 * 128 + 115000 * 64 + 110000 * 128 + 118000 * 64 + 2000 * 32)
    b._PMC_CACHE["data"] = {"error": "rocprofv3 not found"}
    assert b.pmc_traffic("gemm_f16_kernel<256, 128, 4, 2, 7, 64>")["hbm_bytes_per_launch"] is None
    assert b.cluster_pmc_traffic()[0] is None


def test_default_run_reads_the_committed_counters_and_says_so():
    """Without --pmc no profiler child process runs inside the judged invocation: the counters come from the committed copy of a
    `bench.py --pmc` run (or there is none and traffic is null), and the source is named."""
    b = _bench()
    assert b.PMC_LIVE is False
    data, src = b.pmc_counters()
    if "error" in data:
        assert b.pmc_traffic("gemm_f16_kernel<256, 128, 4, 2, 7, 64>")["hbm_bytes_per_launch"] is None
    else:
        assert src.startswith("profiles/") and "not re-counted" in src
        assert any("gemm_f16_kernel" in k for k in data)


def test_kernel_symbol_names_the_instantiation_the_dispatcher_picks():
    b = _bench()
    assert b.kernel_symbol(9600, 3072, 768, 6) == "gemm_f16_kernel<256, 256, 2, 4, 6, 64>"      # c_fc
    assert b.kernel_symbol(9600, 2304, 768, 5) == "gemm_f16_kernel<256, 192, 2, 4, 5, 64>"      # in_proj
    assert b.kernel_symbol(2400, 768, 3072, 7) == "gemm_f16_kernel<64, 64, 2, 2, 7, 128>"       # c_proj, clustered blocks
    assert b.kernel_symbol(9600, 768, 3072, 7) == "gemm_f16_kernel<256, 128, 4, 2, 7, 64>"      # c_proj: one round of 228 tiles
