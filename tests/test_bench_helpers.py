"""bench.py helpers that read the committed profiles (CPU only): the PMC traffic of the roofline's dominant kernel and
of the token-cluster op must resolve against profiles/*traffic_pmc.json - a kernel renamed without refreshing the
profiles would silently turn `roofline.traffic` into null."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_traffic_resolves_for_the_profiled_gemms():
    b = _bench()
    for kernel, algorithmic in (("gemm_f16_kernel:c_fc", 78.4e6), ("gemm_f16_kernel:c_proj", 137e6)):
        tr = b.pmc_traffic(kernel)
        assert tr is not None, kernel
        assert tr["hbm_bytes_per_launch"] == tr["fetch_bytes"] + tr["write_bytes"]
        assert 0.9 * algorithmic < tr["hbm_bytes_per_launch"] < 4 * algorithmic      # measured >= algorithmic, no wild re-reads
    assert b.pmc_traffic("gemm_f16_kernel:unknown") is None


def test_cluster_pmc_traffic_resolves():
    total = _bench().cluster_pmc_traffic()
    assert total is not None and 36.1e6 < total < 4 * 36.1e6
