"""The core clock (one wave counting s_memtime cycles per 100 MHz tick on a second stream) while a GEMM runs back to back:
hand-written kernel vs vendor library vs idle.  Dev tool, GPU only; builds tools/native/libclock_probe.so if missing.
    python tools/clock_during.py [M N K epilogue]"""
import sys, os, ctypes, subprocess, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import torch
import torch.nn.functional as F
from centerclip_amd import ops

so = os.path.join(HERE, "native", "libclock_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so,
                           os.path.join(HERE, "native", "clock_probe.hip")])
lib = ctypes.CDLL(so)
lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def probe_while(tag, fn, ms=60.0, samples=400, spin=250):
    side = torch.cuda.Stream()
    buf = torch.zeros(2 * samples, dtype=torch.int64, device="cuda")
    if fn is not None:
        for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    if fn is not None:
        for _ in range(300): fn(); n += 1                       # get going before the probe starts
    lib.clock_probe_launch(buf.data_ptr(), samples, spin, side.cuda_stream)
    while (time.time() - t0) * 1e3 < ms and fn is not None:
        for _ in range(100): fn(); n += 1
    torch.cuda.synchronize()
    v = buf.cpu().view(samples, 2).double()
    mhz = v[:, 1] / v[:, 0] * 100.0
    dur = float(v[:, 0].sum()) / 100.0
    k = samples // 4
    print(f"{tag:14s} clock over {dur / 1e3:6.1f} ms of probing: median {float(mhz.median()):6.0f} MHz, first quarter {float(mhz[:k].median()):6.0f}, "
          f"last quarter {float(mhz[-k:].median()):6.0f}, min {float(mhz.min()):6.0f}, max {float(mhz.max()):6.0f}  ({n} launches)", flush=True)


def main():
    M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (9600, 3072, 768)
    epi = sys.argv[4] if len(sys.argv) > 4 else "f16"
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    b = torch.randn(N, device="cuda"); bh = b.half()
    out = torch.zeros(M, N, device="cuda", dtype=torch.float16 if epi.startswith("f16") else torch.float32)
    print(f"{M}x{N}x{K} {epi}")
    probe_while("idle", None)
    probe_while("hand-written", lambda: ops.linear_f16(a, w, b, epi, out=out))
    probe_while("library", lambda: F.linear(a, w, bh))
    probe_while("hand-written", lambda: ops.linear_f16(a, w, b, epi, out=out))
    probe_while("library", lambda: F.linear(a, w, bh))


if __name__ == "__main__":
    main()
