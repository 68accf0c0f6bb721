"""End-to-end retrieval evaluation on synthetic data, written the way the reference's main.py drives it (:98, :232-233, :381-499):

    model = CLIP4Clip.from_pretrained(...); eval_epoch(model, test_dataloader, device)

with random-init CLIP weights (no checkpoint / dataset access here) and a synthetic "dataset" of N clips + N captions.
Every compute step runs in the HIP library; swap the state dict for a real ViT-B/32 checkpoint and the loader for
dataloaders/* to evaluate a trained model.

    python examples/eval_synthetic.py [--clips 64] [--algo kmediods++|spectral|pooling]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerclip_amd.clip4clip import CLIP4Clip              # noqa: E402
from centerclip_amd.eval import eval_epoch                  # noqa: E402
import bench                                                # noqa: E402  (cfg-2 task config, random ViT-B/32 state dict)


class SyntheticRetrieval(torch.utils.data.Dataset):
    """(input_ids, input_mask, segment_ids, video, video_mask) as dataloaders/* yield them (main.py:427)."""

    def __init__(self, n, frames=12, words=32, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.video = torch.randn(n, 1, frames, 3, 224, 224, generator=g)
        self.ids = torch.randint(1, 49405, (n, words), generator=g)
        self.ids[:, 0] = 49406
        eot = torch.randint(3, words, (n,), generator=g)
        for i in range(n):
            self.ids[i, eot[i]] = 49407
            self.ids[i, eot[i] + 1:] = 0
        self.mask = (self.ids != 0).long()
        self.vmask = torch.ones(n, 1, frames, dtype=torch.long)

    def __len__(self):
        return self.video.shape[0]

    def __getitem__(self, i):
        return self.ids[i], self.mask[i], torch.zeros_like(self.ids[i]), self.video[i], self.vmask[i]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--algo", default="kmediods++", choices=["kmediods++", "spectral", "pooling", "sparse_sampling"])
    ap.add_argument("--in-flight", type=int, default=None, help="batches in flight (model instances / streams); default: eval_epoch's own (2 on a GPU)")
    a = ap.parse_args()
    device = torch.device("cuda:0")
    c = bench.CFG2
    args = bench.task_config(c)                              # cfg 2: 12 frames -> 3 segments at block 7, K = 49
    args.cluster_algo = a.algo
    vars(args).update(spectral_sigma=2.0, spectral_graph="HeatKernel", spectral_knn_k=1, spectral_spg=0, svd_correct_sign=1)
    model = CLIP4Clip.from_state_dict(bench.random_state_dict(c, seed=0), args).to(device).eval()
    loader = torch.utils.data.DataLoader(SyntheticRetrieval(a.clips), batch_size=a.batch, shuffle=False)
    r1, seconds, info = eval_epoch(model, loader, device, args, log=print, in_flight=a.in_flight)
    print("\n".join(info))
    print("R@1 %.1f (random weights: chance level is %.1f); model time %.2f s for %d clips" % (r1, 100.0 / a.clips, seconds, a.clips))


if __name__ == "__main__":
    main()
