"""Caller glue mirrored from the reference's modelutils.py:1-16 (DEV constant + find_layers).  On ROCm
`cuda:0` is the HIP device, so the reference drivers' `DEV` works unchanged."""
import torch
import torch.nn as nn

DEV = torch.device('cuda:0')


def find_layers(module, layers=(nn.Conv2d, nn.Linear), name=''):
    """{dotted name: module} for every leaf whose exact type is in `layers`, in named_children DFS order
    (the order opt_sequential quantises them in, opt.py:97-170)."""
    if type(module) in tuple(layers):
        return {name: module}
    found = {}
    for child_name, child in module.named_children():
        found.update(find_layers(child, layers=layers, name=f"{name}.{child_name}" if name else child_name))
    return found
