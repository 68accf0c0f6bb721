import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in (15, 30, 59, 118, 472):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, device="cuda", dtype=torch.float16); y = torch.empty_like(x)
    t1 = timeit(lambda: x.zero_()); t2 = timeit(lambda: y.copy_(x))
    print(f"{mb:4d} MB  fill {t1*1e3:7.1f} us {mb/1024/t1*1e3/1e3*1e3:6.2f} TB/s(w)   copy {t2*1e3:7.1f} us {2*mb/1024/t2:6.2f} TB/s(r+w)")
