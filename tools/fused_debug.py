"""dev aid: where does cc_inproj_attention_f16 differ from the two-launch form?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_r4_gpu import _inproj_inputs
for nseq, L, heads, causal in [(192, 50, 12, False), (5, 50, 1, False), (16, 32, 8, True), (3, 17, 2, True)]:
    W = heads * 64
    h16, st, wf, c1, c2 = _inproj_inputs(nseq * L, W, 100 + nseq)
    qkv = ops.linear_ln_f16(h16, wf, c1, c2, st, 1)
    want = ops.attention_f16(qkv, nseq, L, heads, causal=causal)
    got = ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, nseq, L, heads, causal=causal)
    torch.cuda.synchronize()
    bad = (got != want)
    print(nseq, L, heads, "mismatches", int(bad.sum()), "of", bad.numel(), "max", float((got.float() - want.float()).abs().max()))
    if bad.any():
        idx = bad.nonzero()
        rows, cols = idx[:, 0], idx[:, 1]
        print("  tokens:", torch.bincount(rows % L, minlength=L).tolist())
        print("  seq%5:", torch.bincount((rows // L) % 5, minlength=5).tolist())
        print("  d%16:", torch.bincount(cols % 16, minlength=16).tolist())
        print("  head:", torch.bincount(cols // 64, minlength=heads).tolist())
import ctypes
from centerclip_amd import _lib as L
lib = L.lib()
if hasattr(lib, "cc_debug_set_attn_dump"):
    lib.cc_debug_set_attn_dump.argtypes = [ctypes.c_void_p]
    for nseq, L_, heads, causal in [(192, 50, 12, False), (5, 50, 1, False)]:
        W = heads * 64
        h16, st, wf, c1, c2 = _inproj_inputs(nseq * L_, W, 100 + nseq)
        qkv = ops.linear_ln_f16(h16, wf, c1, c2, st, 1)
        dump = torch.zeros_like(qkv)
        lib.cc_debug_set_attn_dump(dump.data_ptr())
        got = ops.inproj_attention_f16(h16, wf, c1, c2, st, 1, nseq, L_, heads, causal=causal)
        torch.cuda.synchronize()
        lib.cc_debug_set_attn_dump(None)
        bad = dump != qkv
        print("qkv dump mismatches", int(bad.sum()), "of", bad.numel(), "q/k/v:", [int(bad[:, i * W:(i + 1) * W].sum()) for i in range(3)])
        want = ops.attention_f16(dump, nseq, L_, heads, causal=causal)
        print("attention on the dumped qkv vs fused:", int((want != got).sum()))
