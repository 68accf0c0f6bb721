"""A few training steps on synthetic data, written the way the reference's main.py drives them (:98-170, :291-378):

    model = CLIP4Clip.from_pretrained(...); optimizer = BertAdam(prep_optim_params_groups(...)); train_epoch(...)

with random-init ViT-B/32 weights at the cfg-2 shape (12 frames -> 3 segments at block 7, K = 49, batch 16) and fp32 master
weights.  Forward and backward of the towers, the loss and the optimizer step run in the HIP library (centerclip_amd.train);
the path is a correctness slice - per-op launches from Python, nothing fused or tuned - and the printed step time says so.

    python examples/train_synthetic.py [--steps 4] [--batch 16]
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/train_synthetic.py   (RCCL, bucketed)
"""
import argparse
import os
import sys
import time
from argparse import Namespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from centerclip_amd.clip4clip import CLIP4Clip              # noqa: E402
from centerclip_amd.train import BertAdam, prep_optim_params_groups, train_epoch   # noqa: E402
from centerclip_amd import dist as ccdist                   # noqa: E402
import bench                                                # noqa: E402
from eval_synthetic import SyntheticRetrieval               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--graph", type=int, default=1, help="also time the step captured into a hipGraph")
    ap.add_argument("--b16", type=int, default=0, help="ViT-B/16 instead (cfg-5 shape: 197 tokens per frame, 12 frames -> 4 segments, "
                                                       "K = 100; the attention backward's two-launch form)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", rank)
    torch.cuda.set_device(device)
    if world > 1:
        torch.distributed.init_process_group("nccl")
    c = bench.CFG2
    if a.b16:
        c = dict(c, name="cfg5-shaped: ViT-B/16", patch=16, T_new=4, K=100)
    args = bench.task_config(c)
    model = CLIP4Clip.from_state_dict(bench.random_state_dict(c, seed=0), args).float().to(device)
    targs = Namespace(lr=1e-7, wd=0.2, new_added_modules=["Cross", "cluster_embed"], gradient_accumulation_steps=1,
                      clip_grad_norm=None)
    opt = BertAdam(prep_optim_params_groups(targs, model, coef_lr=1e-3), lr=targs.lr, warmup=0.1, t_total=100 * a.steps,
                   schedule='warmup_cosine', b1=0.9, b2=0.98, e=1e-6, max_grad_norm=1.0)
    buckets = ccdist.GradientBuckets(model.parameters()) if world > 1 else None
    data = SyntheticRetrieval(a.batch * a.steps, seed=rank)
    loader = torch.utils.data.DataLoader(data, batch_size=a.batch, shuffle=False)
    t = [time.time()]

    def log(epoch, step, loss, sim_loss, gs):
        torch.cuda.synchronize()
        t.append(time.time())
        if rank == 0:
            print("step %d  loss %.4f  %.0f ms" % (gs, loss, (t[-1] - t[-2]) * 1e3), flush=True)
    train_epoch(0, targs, model, loader, device, opt, 0, buckets=buckets, log=log)
    if rank == 0:
        steady = (t[-1] - t[2]) / max(len(t) - 3, 1) if len(t) > 3 else float("nan")
        print("steady step %.0f ms = %.1f clips/s per rank (unfused per-op training path, launched op by op)" % (steady * 1e3, a.batch / steady))
    if a.graph and world == 1:
        # the same step (forward, backward, optimizer, clamp) as ONE hipGraph on static input buffers (train.GraphedTrainStep):
        # no op of it synchronises with the host, so what remains is the GPU time of the unfused kernels
        from centerclip_amd.train import GraphedTrainStep
        gopt = BertAdam(prep_optim_params_groups(targs, model, coef_lr=1e-3), lr=targs.lr, warmup=0.1, t_total=100 * a.steps,
                        schedule='warmup_cosine', b1=0.9, b2=0.98, e=1e-6, max_grad_norm=1.0, capturable=True)
        stepper = GraphedTrainStep(model, gopt)
        batch = next(iter(loader))
        gloss = stepper(batch)
        for _ in range(3):
            stepper(batch)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            gloss = stepper(batch)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / 10 * 1e3
        print("captured step: %.1f ms = %.0f clips/s (loss %.4f)" % (ms, a.batch / ms * 1e3, float(gloss)))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
