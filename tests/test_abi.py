"""CPU-only: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and
exports every symbol include/centerclip_hip.h declares.  No compute calls here."""
import ctypes
import os
import re
from argparse import Namespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from centerclip_amd import build
    return build.build(verbose=False)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "centerclip_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for must in ("cc_version", "cc_token_cluster_f32", "cc_batch_kmedoids_f32", "cc_kmedoids_from_dist_f32",
                 "cc_pairwise_distance_f32"):
        assert must in syms


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    for s in declared_symbols():
        assert hasattr(lib, s), "missing export: " + s
    lib.cc_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.cc_version()
    lib.cc_status_string.restype = ctypes.c_char_p
    assert lib.cc_status_string(-1) == b"invalid argument"


def test_library_exports_nothing_the_header_does_not_declare(libpath):
    """Every cc_* symbol the shipped library exports is declared in include/centerclip_hip.h: no undeclared entry points,
    no debug hooks with process-wide state outside the header's diagnostics section (development builds add theirs with
    -DCC_DEV_KNOBS)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("cc_")
                       and ln.split()[-2] in ("T", "t", "W")})
    assert exported, "nm found no cc_ exports"
    undeclared = [s for s in exported if s not in set(declared_symbols())]
    assert not undeclared, "exported but not declared in the header: %s" % undeclared


def test_workspace_query_and_argument_validation_need_no_gpu(libpath):
    from centerclip_amd import _lib as L
    lib = L.lib()
    assert lib.cc_cluster_workspace_bytes(48, 196, 768, 0) >= 48 * 196 * 196 * 4
    assert lib.cc_cluster_workspace_bytes(48, 196, 768, 1) >= 48 * 196 * (196 + 768) * 4
    assert lib.cc_cluster_workspace_bytes(0, 196, 768, 0) == 0
    # NULL pointers are rejected before anything touches the device
    lay = L.TokenLayout(1, 1, 1, 8, 8 * 16, 0, 0, 16)
    rc = lib.cc_batch_kmedoids_f32(None, ctypes.byref(lay), 16, 2, 0, 2.0, 1e-6, 10, 1, 1, 0, None, None, None, None, 0, None)
    assert rc == -1


def test_product_path_has_no_cpu_fallback():
    import centerclip_amd.cluster as cl
    with pytest.raises(RuntimeError):
        cl.batch_fast_kmedoids(torch.zeros(1, 8, 16), 2)
    with pytest.raises(RuntimeError):
        cl.pairwise_distance(torch.zeros(8, 16), torch.zeros(8, 16))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "centerclip_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


def _args(**kw):
    base = dict(cluster_inter=1, cluster_algo='kmediods++', max_frames=12,
                cluster_num_blocks=[49] * 12, target_frames_blocks=[12] * 6 + [3] * 6,
                cluster_distance='euclidean', cluster_threshold=1e-6, cluster_iter_limit=100,
                minkowski_norm_p=2.0, pretrained_clip_name='ViT-B/32', aggregation=None, pre_norm=False)
    base.update(kw)
    return Namespace(**base)


def test_get_cluster_inter_fires_where_the_reference_does():
    """modules/cluster/cluster.py:23-37: a block clusters iff frames shrink or cluster_num shrinks."""
    from centerclip_amd.cluster import get_cluster_inter
    fired = [b for b in range(1, 13) if get_cluster_inter(768, b, _args()) is not None]
    assert fired == [7]
    m = get_cluster_inter(768, 7, _args())
    assert (m.before_block_frames, m.after_block_frames, m.frame_duration, m.cluster_num, m.split_size) == (12, 3, 4, 49, 16)
    assert get_cluster_inter(768, 7, _args(cluster_inter=0)) is None
    assert get_cluster_inter(768, 7, None) is None
    # tokens shrink while frames stay (cfg-1 style): 49 -> 25 at block 4
    a = _args(target_frames_blocks=[12] * 12, cluster_num_blocks=[49] * 3 + [25] * 9)
    assert [b for b in range(1, 13) if get_cluster_inter(768, b, a) is not None] == [4]
    assert get_cluster_inter(768, 7, _args(pretrained_clip_name='ViT-B/16')).split_size == 4
