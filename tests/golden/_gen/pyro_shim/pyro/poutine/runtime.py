"""Handler stack + default site behaviour (shim; see package docstring in pyro/__init__.py)."""
import torch

_STACK = []
EPS_LOG = []      # every reparameterised noise draw, in program order (captured into fixtures)
IDX_LOG = []      # every subsample index draw
DISC_LOG = []     # every draw of a non-reparameterised (discrete) site that was actually sampled (guide sites)


def apply_stack(msg):
    # innermost handler first, like pyro.poutine.runtime.apply_stack
    for frame in reversed(_STACK):
        frame._process_message(msg)
    if msg["type"] == "sample" and msg["value"] is None:
        fn = msg["fn"]
        if getattr(fn, "has_rsample", False):
            msg["value"] = fn.rsample()
        else:
            msg["value"] = fn.sample()
            DISC_LOG.append(msg["value"].detach().clone())
    elif msg["type"] == "subsample" and msg["value"] is None:
        size, ss = msg["size"], msg["subsample_size"]
        if ss is None or ss >= size:
            msg["value"] = torch.arange(size)
        else:
            msg["value"] = torch.randperm(size)[:ss]
        IDX_LOG.append(msg["value"].clone())
    for frame in _STACK:
        frame._postprocess_message(msg)
    return msg
