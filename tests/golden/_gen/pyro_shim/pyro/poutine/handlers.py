import collections

import torch

from .messenger import Messenger


class Trace(object):
    def __init__(self):
        self.nodes = collections.OrderedDict()


class TraceMessenger(Messenger):
    def __init__(self, param_only=False):
        super().__init__()
        self.param_only = param_only
        self.trace = Trace()

    def __enter__(self):
        self.trace = Trace()
        return super().__enter__()

    def _postprocess_message(self, msg):
        if self.param_only and msg["type"] != "param":
            return
        if msg["type"] in ("param", "sample", "subsample"):
            key = msg["name"] if msg["type"] != "subsample" else "__plate__" + msg["name"]
            self.trace.nodes[key] = dict(msg)


def trace(fn=None, param_only=False):
    assert fn is None
    return TraceMessenger(param_only=param_only)


class ReplayMessenger(Messenger):
    """Replays guide sample sites and plate indices into the model."""

    def __init__(self, guide_trace):
        super().__init__()
        self.guide_trace = guide_trace

    def _process_message(self, msg):
        if msg["type"] == "sample" and not msg["is_observed"] and msg["name"] in self.guide_trace.nodes:
            msg["value"] = self.guide_trace.nodes[msg["name"]]["value"]
            msg["replayed"] = True
        elif msg["type"] == "subsample":
            key = "__plate__" + msg["name"]
            if key in self.guide_trace.nodes:
                msg["value"] = self.guide_trace.nodes[key]["value"]


class EnumMessenger(Messenger):
    """Parallel enumeration of model-side discrete sites absent from the guide: the value is
    the full support with the enumeration dim placed left of the (single) plate dim."""

    def __init__(self, first_available_dim=-2):
        super().__init__()
        self.dim = first_available_dim

    def _process_message(self, msg):
        if (msg["type"] == "sample" and not msg["is_observed"] and msg["value"] is None
                and msg["infer"].get("enumerate") == "parallel"):
            support = msg["fn"].enumerate_support(expand=False)   # (C, 1, ...) batch dims kept as 1
            c = support.shape[0]
            msg["value"] = support.reshape((c,) + (1,) * (-self.dim - 1))
            msg["infer"]["_enumerated"] = True


class EnumConfigMessenger(Messenger):
    def _process_message(self, msg):
        if (msg["type"] == "sample" and not msg["is_observed"]
                and getattr(msg["fn"], "has_enumerate_support", False)):
            msg["infer"].setdefault("enumerate", "parallel")
