"""Clock / power while one GEMM shape runs back to back: the hand-written kernel and the vendor library (dev tool, GPU only).
    python tools/lib_power.py [M N K epilogue]"""
import sys, os, subprocess, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from centerclip_amd import ops


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
            pw = re.findall(r"Power \(W\): ([\d.]+)", txt)
            if sclk and pw: out.append((int(sclk[0]), float(pw[0])))
        except Exception:
            pass
        time.sleep(0.2)


def run(tag, fn, seconds=6.0):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); e0.record()
    while time.time() - t0 < seconds:
        for _ in range(200): fn()
        n += 200
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    out = out[2:] if len(out) > 4 else out
    clk = sum(o[0] for o in out) / max(len(out), 1); pw = sum(o[1] for o in out) / max(len(out), 1)
    print(f"{tag:12s} {us:7.1f} us / launch (back to back)   sclk {clk:6.0f} MHz   power {pw:5.0f} W   ({len(out)} samples)", flush=True)


def main():
    M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (9600, 3072, 768)
    epi = sys.argv[4] if len(sys.argv) > 4 else "f16_gelu"
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    b = torch.randn(N, device="cuda"); bh = b.half()
    out = torch.zeros(M, N, device="cuda", dtype=torch.float16 if epi.startswith("f16") else torch.float32)
    print(f"{M}x{N}x{K} {epi}")
    run("hand-written", lambda: ops.linear_f16(a, w, b, epi, out=out))
    run("library", lambda: F.linear(a, w, bh))
    run("hand-written", lambda: ops.linear_f16(a, w, b, epi, out=out))
    run("library", lambda: F.linear(a, w, bh))


if __name__ == "__main__":
    main()
