"""Dev probe (GPU only): the cost of a stream-K fix-up exchange between workgroups - see tools/native/sk_probe.hip."""
import sys, os, ctypes, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
import torch
so = os.path.join(HERE, "native", "libsk_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "native", "sk_probe.hip")])
lib = ctypes.CDLL(so)
lib.sk_probe_launch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
W = 512
part = torch.zeros(W * 16 * 256 * 4, device="cuda"); out = torch.zeros_like(part)
flag = torch.zeros(W + 16, dtype=torch.int32, device="cuda"); err = torch.zeros(2, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
epoch = [0]
def run(mode, dist, work, n=200):
    def one():
        epoch[0] += 1
        lib.sk_probe_launch(part.data_ptr(), flag.data_ptr(), out.data_ptr(), err.data_ptr(), mode, dist, epoch[0], work, W, st)
    for _ in range(10): one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): one()
    e1.record(); torch.cuda.synchronize()
    bad = err.cpu().tolist(); err.zero_()
    return e0.elapsed_time(e1) / n * 1e3, bad
for work in (0, 20):
    base, _ = run(0, 8, work)
    line = f"k-loop stand-in {work:2d}: no exchange {base:6.1f} us"
    for mode, name in ((1, "release / acquire fences"), (2, "vmcnt(0) + acquire")):
        for dist, where in ((8, "same XCD"), (1, "next XCD")):
            t, bad = run(mode, dist, work)
            line += f" | {name}, {where}: {t:6.1f} (give-ups {bad[0]}, stale reads {bad[1]})"
    print(line, flush=True)
