"""Round-6 parity cases (need a real MI355X, ``-m gpu``).

  * N3 through the strip form of the uint8 patch gather (csrc/transformer.hip im2col_u8_strip_kernel: shifts instead of
    divisions, every CLIP patch size): bit-identical patch matrices / features to the float loader path at ViT-B/32 and
    ViT-B/16 geometry, and `linear_patch='3d'` from uint8 frames (was CC_ERR_UNSUPPORTED), dataloaders/transforms.py:19-34,166
    + modules/clip.py:296-317.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g2():
    return np.load(os.path.join(HERE, "golden", "r2_golden.npz"))


def _clip(g2, linear_patch, patch=None):
    from centerclip_amd.clip import CLIP
    E, RES, P, VW, VL, CTX, VOCAB, TW, TH, TL, B, T, T_new = [int(v) for v in g2["s1_cfg"]]
    sd = {k[6:]: torch.from_numpy(g2[k].astype(np.float32) if g2[k].dtype == np.float16 else g2[k])
          for k in g2.files if k.startswith("s1_sd/")}
    model = CLIP(E, RES, VL, VW, patch or P, CTX, VOCAB, TW, TH, TL, linear_patch=linear_patch, video_frames=T, args=None)
    if patch is None or patch == P:
        model.load_state_dict(sd, strict=False)
    return model, RES, T


@pytest.mark.parametrize("linear_patch", ["2d", "3d"])
@pytest.mark.parametrize("patch", [None, 8])
def test_uint8_frames_strip_gather_bit_identical(g2, linear_patch, patch):
    """uint8 frames (CHW and the decoder's HWC) == the loader's float path, bit for bit, for both patch embeddings; with a
    second patch size (8: 3 shift amounts differ) on randomly initialised weights."""
    from oracle import clip_oracle as clo
    torch.manual_seed(3)
    model, RES, T = _clip(g2, linear_patch, patch)
    if linear_patch == "3d":
        with torch.no_grad():
            model.visual.conv2.weight.normal_(0, 0.02)
    model = model.to(DEV).eval()
    rng = np.random.default_rng(17)
    u_hwc = torch.from_numpy(rng.integers(0, 256, size=(3 * T, RES, RES, 3), dtype=np.uint8))
    u_chw = u_hwc.permute(0, 3, 1, 2).contiguous()
    x = clo.loader_normalize(u_hwc, channels_last=True)
    with torch.no_grad():
        f_ref, _ = model.encode_image(x.to(DEV), video_frame=T)
        f_chw, _ = model.encode_image(u_chw.to(DEV), video_frame=T)
        f_hwc, _ = model.encode_image(u_hwc.to(DEV), video_frame=T)
    assert bool(torch.isfinite(f_ref).all())
    assert torch.equal(f_chw, f_ref) and torch.equal(f_hwc, f_ref)
