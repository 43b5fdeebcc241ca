"""Test stand-in for `hloc/utils/base_model.py` (the plugin protocol of the reference's hloc, :8-47), restated for the
tests of `gim_amd/hloc_matchers/`: a matcher is an nn.Module whose constructor merges the user's conf over the class's
`default_conf` and hands the result to `_init`, and whose `forward` refuses a data dict that lacks one of
`required_inputs` before delegating to `_forward`.  `dynamic_load(root, name)` returns the one class of module
`root.<name>` that derives from `BaseModel`."""
import importlib
import inspect

import torch


class BaseModel(torch.nn.Module):
    default_conf = {}
    required_inputs = []

    def __init__(self, conf):
        torch.nn.Module.__init__(self)
        merged = dict(self.default_conf)
        merged.update(conf)
        self.conf = merged
        self.required_inputs = list(self.required_inputs)
        self._init(merged)

    def forward(self, data):
        missing = [k for k in self.required_inputs if k not in data]
        assert not missing, "Missing key {} in data".format(missing[0])
        return self._forward(data)

    def _init(self, conf):
        raise NotImplementedError(type(self).__name__ + "._init")

    def _forward(self, data):
        raise NotImplementedError(type(self).__name__ + "._forward")


def dynamic_load(root, model):
    name = root.__name__ + "." + model
    mod = importlib.import_module(name)
    found = [c for _, c in inspect.getmembers(mod, inspect.isclass) if c.__module__ == name and issubclass(c, BaseModel)]
    assert len(found) == 1, found
    return found[0]
