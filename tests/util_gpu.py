"""Helpers shared by the GPU tests: bf16 tensors, Mat views, error metrics."""
import numpy as np
import torch

from zero_amd.func import Engine, Mat

_ENG = None


def eng():
    global _ENG
    if _ENG is None:
        _ENG = Engine("cuda:0")
    return _ENG


def bf(x):
    return x.to(torch.bfloat16).contiguous()


def rand_bf(*shape, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    if seed is not None:
        g.manual_seed(seed)
    return bf((torch.randn(*shape, generator=g) * scale)).cuda()


def mat(t, rows=None, cols=None, ld=None, off=0):
    if rows is None:
        rows, cols = t.shape[0], t.shape[1]
    return Mat(t, rows, cols, ld if ld is not None else t.shape[-1], off)


def rel_err(a, b):
    a = a.detach().float().cpu().double()
    b = b.detach().float().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())
