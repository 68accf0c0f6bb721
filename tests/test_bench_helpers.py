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
    for kernel, algorithmic in (("gemm_f16_kernel<256, 256, 2, 4, 6, 64, 1>", 78.4e6), ("gemm_f16_kernel<256, 128, 4, 2, 7, 64, 1>", 113e6)):
        tr = b.pmc_traffic(kernel)
        assert tr is not None, kernel
        assert abs(tr["hbm_bytes_per_launch"] - (tr["fetch_bytes"] + tr["write_bytes"])) <= 2          # (each rounded)
        assert 0.9 * algorithmic < tr["hbm_bytes_per_launch"] < 4 * algorithmic      # measured >= algorithmic, no wild re-reads
    assert b.pmc_traffic("gemm_f16_kernel<1, 2, 3, 4, 5, 6>") is None


def test_kernel_symbol_names_the_instantiation_the_dispatcher_picks():
    b = _bench()
    assert b.kernel_symbol(9600, 3072, 768, 6) == "gemm_f16_kernel<256, 256, 2, 4, 6, 64, 1>"      # c_fc
    assert b.kernel_symbol(9600, 2304, 768, 5) == "gemm_f16_kernel<256, 192, 2, 4, 5, 64, 1>"      # in_proj
    assert b.kernel_symbol(2400, 768, 3072, 7) == "gemm_f16_kernel<64, 64, 2, 2, 7, 128, 1>"       # c_proj, clustered blocks
    assert b.kernel_symbol(9600, 768, 3072, 7) == "gemm_f16_kernel<256, 128, 4, 2, 7, 64, 1>"      # c_proj: one round of 228 tiles


def test_cluster_pmc_traffic_resolves():
    total = _bench().cluster_pmc_traffic()
    assert total is not None and 36.1e6 < total < 4 * 36.1e6
