import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops
def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, N, epi) in [(9600, 3072, "f16_gelu"), (9600, 2304, "f16"), (9600, 768, "f32_resid"), (2400, 2304, "f16"), (2400, 768, "f32_resid")]:
    for tile in (1, 5, 4):
        line = f"M={M} N={N} {epi} tile{tile}:"
        for K in (64, 384, 768, 1536, 3072):
            a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
            b = torch.randn(N, device="cuda")
            out = torch.zeros(M, N, device="cuda", dtype=torch.float16 if epi.startswith("f16") else torch.float32)
            try:
                ms = timeit(lambda: ops.linear_f16(a, w, b, epi, out=out, tile=tile))
                line += f"  K={K}: {ms*1e3:6.1f}us"
            except Exception as e:
                line += f"  K={K}: n/a"
        print(line, flush=True)
