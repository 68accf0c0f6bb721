"""bench.py helpers (CPU only): the byte arithmetic of the live counter passes that fill `roofline.traffic` (round 5: measured by
the invocation itself, bench.pmc_counters_live), the name matching of kernel symbols against rocprofv3's kernel names, and the
dispatcher's tile -> symbol mapping."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_traffic_from_live_counters():
    """The calibration launches of profiles/r05_traffic_reconcile.txt as counter records: a 512 MiB copy (4,194,510 read requests
    of 128 B, 8,388,608 write requests of 64 B) and a GEMM symbol; a failed counter pass yields null + the reason, never a
    number from somewhere else."""
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
    assert tr["hbm_bytes_per_launch"] == tr["fetch_bytes"] + tr["write_bytes"] and "measured in this run" in tr["source"]
    assert b.pmc_traffic("gemm_f16_kernel<1, 2, 3, 4, 5, 6>")["hbm_bytes_per_launch"] is None
    total = b.cluster_pmc_traffic()
    assert total == round(230000 * 128 + 115000 * 64 + 110000 * 128 + 118000 * 64 + 2000 * 32)
    b._PMC_CACHE["data"] = {"error": "rocprofv3 not found"}
    assert b.pmc_traffic("gemm_f16_kernel<256, 128, 4, 2, 7, 64>")["hbm_bytes_per_launch"] is None
    assert b.cluster_pmc_traffic() is None


def test_kernel_symbol_names_the_instantiation_the_dispatcher_picks():
    b = _bench()
    assert b.kernel_symbol(9600, 3072, 768, 6) == "gemm_f16_kernel<256, 256, 2, 4, 6, 64>"      # c_fc
    assert b.kernel_symbol(9600, 2304, 768, 5) == "gemm_f16_kernel<256, 192, 2, 4, 5, 64>"      # in_proj
    assert b.kernel_symbol(2400, 768, 3072, 7) == "gemm_f16_kernel<64, 64, 2, 2, 7, 128>"       # c_proj, clustered blocks
    assert b.kernel_symbol(9600, 768, 3072, 7) == "gemm_f16_kernel<256, 128, 4, 2, 7, 64>"      # c_proj: one round of 228 tiles
