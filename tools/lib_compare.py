"""The step's GEMM shapes: the hand-written fp16 kernel (default tile choice, fused epilogue) beside the vendor library
(torch.nn.functional.linear on fp16 = hipBLASLt / rocBLAS, plain epilogue: bias only, fp16 out).  Dev tool, GPU only.
The library leg is a yardstick for what the matrix pipeline gives at these shapes; it is not linked by the product.

    python tools/lib_compare.py [iters]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from centerclip_amd import ops


def timeit(fn, iters, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    shapes = [(9600, 2304, 768, "f16", "in_proj"), (9600, 768, 768, "f32_resid", "out_proj"), (9600, 3072, 768, "f16_gelu", "c_fc"),
              (9600, 768, 3072, "f32_resid", "c_proj"), (2400, 2304, 768, "f16", "in_proj (clustered)"),
              (2400, 768, 768, "f32_resid", "out_proj (clustered)"), (2400, 3072, 768, "f16_gelu", "c_fc (clustered)"),
              (2400, 768, 3072, "f32_resid", "c_proj (clustered)"), (9408, 768, 3072, "f32", "patch_embed"),
              (4096, 4096, 4096, "f16", "square 4k"), (8192, 8192, 8192, "f16", "square 8k")]
    print(f"{'shape':>20s} {'role':22s} | {'ours us':>8s} {'TF':>6s} | {'lib us':>8s} {'TF':>6s} | lib/ours")
    for M, N, K, epi, role in shapes:
        a = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        b = torch.randn(N, device="cuda")
        bh = b.half()
        out = torch.zeros(M, N, device="cuda", dtype=torch.float16 if epi.startswith("f16") else torch.float32)
        t_own = timeit(lambda: ops.linear_f16(a, w, b, epi, out=out), iters)
        t_lib = timeit(lambda: F.linear(a, w, bh), iters)
        fl = 2.0 * M * N * K
        print(f"{M:6d}x{N:5d}x{K:5d} {role:22s} | {t_own:8.1f} {fl/t_own/1e6:6.0f} | {t_lib:8.1f} {fl/t_lib/1e6:6.0f} | {t_lib/t_own:5.2f}",
              flush=True)


if __name__ == "__main__":
    main()
