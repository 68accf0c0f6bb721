"""Host-side logic of the training path (no GPU): the learning-rate schedules and the parameter groups of
utils/optimization.py, BertAdam's argument validation and lr bookkeeping, train_epoch's control flow with a stand-in model."""
import math
from argparse import Namespace

import pytest
import torch

from centerclip_amd import train as cctrain


def test_schedules_are_the_references():
    # utils/optimization.py:25-52: values written out from the definitions
    assert cctrain.warmup_linear(0.05, 0.1) == pytest.approx(0.5) and cctrain.warmup_linear(0.1, 0.1) == pytest.approx(1.0)
    assert cctrain.warmup_linear(0.55, 0.1) == pytest.approx((0.55 - 1.0) / (0.1 - 1.0)) and cctrain.warmup_linear(1.2, 0.1) == 0
    assert cctrain.warmup_constant(0.01, 0.1) == pytest.approx(0.1) and cctrain.warmup_constant(0.5, 0.1) == 1.0
    assert cctrain.warmup_cosine(0.02, 0.1) == pytest.approx(0.2)
    assert cctrain.warmup_cosine(0.5, 0.1) == pytest.approx(0.5 * (1.0 + math.cos(math.pi * 0.5)))
    assert set(cctrain.SCHEDULES) == {"warmup_cosine", "warmup_constant", "warmup_linear"}


def test_bertadam_arguments_and_lr_bookkeeping():
    p = torch.nn.Parameter(torch.zeros(3))
    for bad in (dict(lr=-1.0), dict(lr=1e-3, schedule="nope"), dict(lr=1e-3, warmup=1.5), dict(lr=1e-3, b1=1.0),
                dict(lr=1e-3, b2=-0.1), dict(lr=1e-3, e=-1e-6)):
        with pytest.raises(ValueError):
            cctrain.BertAdam([p], **bad)
    opt = cctrain.BertAdam([{"params": [p], "lr": 2e-3}], lr=1e-3, warmup=0.1, t_total=100, schedule="warmup_linear")
    assert opt.defaults["max_grad_norm"] == 1.0 and opt.defaults["weight_decay"] == 0.01 and opt.defaults["e"] == 1e-6
    assert opt.get_lr() == []                                     # no gradient yet
    p.grad = torch.zeros(3)
    assert opt.get_lr() == [0]                                    # a gradient but no state: the reference returns [0]
    opt.state[p]["step"] = 5
    assert opt.get_lr() == [pytest.approx(2e-3 * cctrain.warmup_linear(5 / 100, 0.1))]
    opt2 = cctrain.BertAdam([p], lr=1e-3)                         # t_total = -1: constant learning rate
    opt2.state[p]["step"] = 7
    assert opt2.get_lr() == [1e-3]


def test_param_groups_follow_the_reference():
    """utils/optimization.py:173-208: CLIP parameters at lr * coef_lr, modules named in new_added_modules at lr, no weight
    decay for biases and LayerNorm parameters."""
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.clip = torch.nn.ModuleDict({"visual": torch.nn.Linear(4, 4), "cluster_embed": torch.nn.Linear(4, 4)})
            self.head = torch.nn.Linear(4, 2)
    m = M()
    args = Namespace(lr=1e-2, wd=0.2, new_added_modules=["cluster_embed"])
    groups = cctrain.prep_optim_params_groups(args, m, coef_lr=0.1)
    ids = lambda g: {id(p) for p in g["params"]}
    assert ids(groups[0]) == {id(m.clip["visual"].weight)} and groups[0]["lr"] == pytest.approx(1e-3) and groups[0]["weight_decay"] == 0.2
    assert ids(groups[1]) == {id(m.clip["visual"].bias)} and groups[1]["weight_decay"] == 0.0 and groups[1]["lr"] == pytest.approx(1e-3)
    assert ids(groups[2]) == {id(m.clip["cluster_embed"].weight), id(m.head.weight)} and "lr" not in groups[2]
    assert ids(groups[3]) == {id(m.clip["cluster_embed"].bias), id(m.head.bias)} and groups[3]["weight_decay"] == 0.0


def test_train_epoch_control_flow():
    """main.py:291-378 with a stand-in model / optimizer: zero_grad -> scheduler -> forward -> backward -> [buckets] -> step every
    gradient_accumulation_steps batches -> logit_scale clamp."""
    calls = []

    class Clip(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.logit_scale = torch.nn.Parameter(torch.tensor(9.0))

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.clip = Clip()
            self.w = torch.nn.Parameter(torch.ones(2))

        def forward(self, ids, seg, mask, video, vmask):
            loss = (self.w * video.float().mean()).sum() + 0 * self.clip.logit_scale
            return {"loss": loss, "sim_loss": loss.detach(), "cluster_loss": torch.zeros(())}

    class Opt:
        def zero_grad(self): calls.append("zero")
        def step(self): calls.append("step")

    class Buckets:
        def reduce(self): calls.append("reduce")
    m = Model()
    batch = tuple(torch.ones(2, 3) for _ in range(5))
    args = Namespace(gradient_accumulation_steps=2, clip_grad_norm=None)
    loss, gs = cctrain.train_epoch(0, args, m, [batch] * 4, "cpu", Opt(), 10, scheduler=lambda o, global_step: calls.append(("sched", global_step)),
                                   buckets=Buckets())
    assert gs == 12 and loss == pytest.approx(1.0)                 # loss / accumulation steps, two optimizer steps
    assert calls == ["zero", ("sched", 10), "zero", ("sched", 10), "reduce", "step", "zero", ("sched", 11), "zero", ("sched", 11),
                     "reduce", "step"]
    assert float(m.clip.logit_scale.detach()) == pytest.approx(4.6052) and m.training


def test_train_epoch_scaler_branch_control_flow():
    """main.py:319-330 (the --fp16 branch): scale(loss).backward -> [reduce] -> unscale_ before clipping -> scaler.step(optimizer)
    -> scaler.update; a step the scaler refuses (inf in a gradient) leaves the parameters alone."""
    calls = []

    class Clip(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.logit_scale = torch.nn.Parameter(torch.tensor(1.0))

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.clip = Clip()
            self.w = torch.nn.Parameter(torch.ones(2))

        def forward(self, ids, seg, mask, video, vmask):
            loss = (self.w * video.float().mean()).sum() + 0 * self.clip.logit_scale
            return {"loss": loss, "sim_loss": loss.detach(), "cluster_loss": torch.zeros(())}

    class Scaler:                                                  # GradScaler's protocol, scale 8, overflow on request
        def __init__(self, overflow_steps=()):
            self.n, self.overflow_steps = 0, set(overflow_steps)
        def scale(self, loss): calls.append("scale"); return loss * 8.0
        def unscale_(self, opt):
            calls.append("unscale")
            for p in m.parameters():
                if p.grad is not None:
                    p.grad.div_(8.0)
        def step(self, opt):
            calls.append("sstep")
            if self.n not in self.overflow_steps:
                opt.step()
            self.n += 1
        def update(self): calls.append("update")

    m = Model()
    opt = torch.optim.SGD(m.parameters(), lr=0.5)
    batch = tuple(torch.ones(2, 3) for _ in range(5))
    args = Namespace(gradient_accumulation_steps=1, clip_grad_norm=10.0)
    loss, gs = cctrain.train_epoch(0, args, m, [batch] * 2, "cpu", opt, 0, scaler=Scaler(overflow_steps=(1,)))
    assert gs == 2 and calls == ["scale", "unscale", "sstep", "update"] * 2
    # first step applied with the UNSCALED gradient (1 per element: w = 1 - 0.5), second one skipped
    assert torch.allclose(m.w.detach(), torch.full((2,), 0.5))
