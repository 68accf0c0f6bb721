"""Which aten / custom ops launch the small kernels of a training step (dev tool): torch.profiler over one eager step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from argparse import Namespace
import torch
from torch.profiler import profile, ProfilerActivity
from centerclip_amd.clip4clip import CLIP4Clip
from centerclip_amd.train import BertAdam, prep_optim_params_groups
import bench
from eval_synthetic import SyntheticRetrieval

c = bench.CFG2
args = bench.task_config(c)
model = CLIP4Clip.from_state_dict(bench.random_state_dict(c, seed=0), args).float().cuda().train()
targs = Namespace(lr=1e-7, wd=0.2, new_added_modules=["Cross", "cluster_embed"], gradient_accumulation_steps=1, clip_grad_norm=None)
opt = BertAdam(prep_optim_params_groups(targs, model, coef_lr=1e-3), lr=targs.lr, warmup=0.1, t_total=100, schedule='warmup_cosine',
               b1=0.9, b2=0.98, e=1e-6, max_grad_norm=1.0)
data = SyntheticRetrieval(16, seed=0)
batch = [t.cuda() for t in next(iter(torch.utils.data.DataLoader(data, batch_size=16)))]


def step():
    opt.zero_grad(set_to_none=True)
    loss = model(*batch)
    if isinstance(loss, (tuple, list)):
        loss = loss[0]
    if isinstance(loss, dict):
        loss = loss["loss"] if "loss" in loss else sum(loss.values())
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=4)
rows = [e for e in ka if e.key in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::copy_", "aten::clone", "aten::zeros_like", "aten::add_", "aten::mul", "aten::add")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print(e.count, e.key, "|", " <- ".join(s.split("/")[-1] for s in e.stack[:4]))
