"""Real-time timeline of the attention launch of a full-size block (dev tool, GPU only): per workgroup (wave 0) entry, operands
staged, exit - 100 MHz stamps through the cc_debug_set_att_profile hook."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops, _lib as L
lib = L.lib()
lib.cc_debug_set_att_profile.argtypes = [ctypes.c_void_p]
for nseq, Ltok, heads in [(192, 50, 12), (48, 50, 12), (16, 32, 8)]:
    W = heads * 64
    qkv = (torch.randn(nseq * Ltok, 3 * W, device="cuda") * 0.5).half()
    for _ in range(10): ops.attention_f16(qkv, nseq, Ltok, heads)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ops.attention_f16(qkv, nseq, Ltok, heads)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    buf = torch.zeros(4096, 4, dtype=torch.long, device="cuda")
    for _ in range(3): ops.attention_f16(qkv, nseq, Ltok, heads)
    lib.cc_debug_set_att_profile(ctypes.c_void_p(buf.data_ptr()))
    ops.attention_f16(qkv, nseq, Ltok, heads)
    torch.cuda.synchronize()
    lib.cc_debug_set_att_profile(ctypes.c_void_p(0))
    t = buf.cpu().double()
    t = t[t[:, 2] > 0] / 100.0
    t0 = t[:, 0].min()
    ent = t[:, 0] - t0; staged = t[:, 1] - t[:, 0]; comp = t[:, 2] - t[:, 1]; ext = t[:, 2] - t0
    print(f"{nseq} x {Ltok} x {heads} heads: {us:5.1f} us / launch back to back, {len(t)} workgroups | entry median {float(ent.median()):.1f} max {float(ent.max()):.1f} | "
          f"load + transpose median {float(staged.median()):.1f} max {float(staged.max()):.1f} | scores / softmax / PV median {float(comp.median()):.1f} max {float(comp.max()):.1f} | "
          f"exit median {float(ext.median()):.1f} last {float(ext.max()):.1f} us", flush=True)
