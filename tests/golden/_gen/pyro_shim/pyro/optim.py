"""pyro.optim subset (shim): one torch optimizer per parameter tensor, per-tensor kwargs through a
callable(module_name, param_name); PyroLRScheduler steps every per-tensor scheduler
(SURVEY.md App. B.6)."""
import torch


def _names(p):
    import pyro
    full = pyro.get_param_store().param_name(p)
    module_name = full.split("$$$")[0]
    stripped = full.split("$$$")[1] if "$$$" in full else full
    return module_name, stripped


class PyroOptim(object):
    def __init__(self, optim_constructor, optim_args):
        self.pt_optim_constructor = optim_constructor
        self.pt_optim_args = optim_args
        self.optim_objs = {}

    def _args(self, p):
        if callable(self.pt_optim_args):
            return dict(self.pt_optim_args(*_names(p)))
        return dict(self.pt_optim_args)

    def __call__(self, params):
        for p in params:
            if id(p) not in self.optim_objs:
                self.optim_objs[id(p)] = self.pt_optim_constructor([p], **self._args(p))
            self.optim_objs[id(p)].step()


class Adam(PyroOptim):
    def __init__(self, optim_args):
        super().__init__(torch.optim.Adam, optim_args)


class PyroLRScheduler(PyroOptim):
    def __init__(self, scheduler_constructor, optim_args):
        self.pt_scheduler_constructor = scheduler_constructor
        pt_optim_constructor = optim_args.pop('optimizer')
        pt_optim_args = optim_args.pop('optim_args')
        self.kwargs = optim_args
        super().__init__(pt_optim_constructor, pt_optim_args)
        self.scheds = {}

    def __call__(self, params):
        for p in params:
            if id(p) not in self.optim_objs:
                opt = self.pt_optim_constructor([p], **self._args(p))
                self.optim_objs[id(p)] = opt
                self.scheds[id(p)] = self.pt_scheduler_constructor(opt, **self.kwargs)
            self.optim_objs[id(p)].step()

    def step(self, *a, **k):
        for s in self.scheds.values():
            s.step()


def MultiStepLR(optim_args):
    return PyroLRScheduler(torch.optim.lr_scheduler.MultiStepLR, optim_args)


def StepLR(optim_args):
    return PyroLRScheduler(torch.optim.lr_scheduler.StepLR, optim_args)
