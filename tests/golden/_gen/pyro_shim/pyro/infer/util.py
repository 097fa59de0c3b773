import torch


def torch_item(x):
    return x if isinstance(x, (int, float)) else x.item()


def zero_grads(tensors):
    for p in tensors:
        if p.grad is not None:
            p.grad = torch.zeros_like(p.grad)
