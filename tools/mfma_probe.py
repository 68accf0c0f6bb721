"""Dev probe (GPU only): sustained fp16 MFMA rate from registers only, 16x16x32 against 32x32x16, with board power and the
core clock read by a probe wave on a second stream (tools/native/clock_probe.hip) - what the matrix pipe gives under the
power limit when nothing else moves."""
import sys, os, ctypes, subprocess, threading, time, re
HERE = os.path.dirname(os.path.abspath(__file__))
import torch
def load(name):
    so = os.path.join(HERE, "native", "lib%s.so" % name)
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "native", name + ".hip")])
    return ctypes.CDLL(so)
lib = load("mfma_probe"); clk = load("clock_probe")
lib.mfma_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
clk.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(256 * 512, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def power():
    try:
        txt = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        m = re.findall(r"Power \(W\): ([\d.]+)", txt)
        return float(m[0]) if m else float("nan")
    except Exception:
        return float("nan")
iters = 20000
for shape in (16, 32):
    for threads in (256, 512):
        flops = 256 * (threads // 64) * iters * 16 * 2 * 16 * 16 * 32        # per launch
        for _ in range(3): lib.mfma_probe_launch(out.data_ptr(), shape, iters, threads, 256, st)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(); buf = torch.zeros(2 * 200, dtype=torch.int64, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 40
        for i in range(n):
            lib.mfma_probe_launch(out.data_ptr(), shape, iters, threads, 256, st)
            if i == 5: clk.clock_probe_launch(buf.data_ptr(), 200, 250, side.cuda_stream)
        pw = power()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        v = buf.cpu().view(200, 2).double(); mhz = float((v[:, 1] / v[:, 0] * 100.0).median())
        cyc = ms * 1e-3 * mhz * 1e6 / (iters * 16)                              # core cycles per 16x16x32-equivalent MFMA per wave
        print(f"v_mfma_f32_{shape}x{shape}x{32 if shape == 16 else 16}_f16, {threads // 64 // 4} wave(s) per SIMD: {flops / ms / 1e9:7.0f} TFLOP/s, "
              f"{ms:6.2f} ms / launch, clock (probe wave) {mhz:5.0f} MHz, board power {pw:5.0f} W, {cyc:5.1f} cycles per 16 KFLOP MFMA issue slot", flush=True)

lib.mfma_lds_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
iters = 4000
for writes in (0, 1):
    flops = 256 * 2.0 * 256 * 256 * 64 * iters
    for _ in range(3): lib.mfma_lds_probe_launch(out.data_ptr(), iters, writes, 256, st)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); buf = torch.zeros(2 * 200, dtype=torch.int64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for i in range(n):
        lib.mfma_lds_probe_launch(out.data_ptr(), iters, writes, 256, st)
        if i == 3: clk.clock_probe_launch(buf.data_ptr(), 200, 250, side.cuda_stream)
    pw = power()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    v = buf.cpu().view(200, 2).double(); mhz = float((v[:, 1] / v[:, 0] * 100.0).median())
    print(f"256x256 tile k-loop from LDS ({'fragment reads + 64 KB of ds_write per k-step' if writes else 'fragment reads only'}, no global traffic): "
          f"{flops / ms / 1e9:7.0f} TFLOP/s, {ms * 1e3 / iters:6.3f} us per k-step, clock (probe wave) {mhz:5.0f} MHz, board power {pw:5.0f} W", flush=True)
