"""`BaseModel` protocol of hloc plugins (hloc/utils/base_model.py:8-33): `default_conf` merged under the given conf,
`required_inputs` asserted by `forward`, subclasses implement `_init(conf)` / `_forward(data)`.  hloc's own class is used when
hloc is importable, so that `hloc.utils.base_model.dynamic_load` accepts the plugin."""
import sys
from abc import ABCMeta, abstractmethod
from copy import copy

from torch import nn

try:  # the reference's hloc on the path: be a genuine subclass of ITS BaseModel
    from hloc.utils.base_model import BaseModel  # noqa: F401
except Exception:  # noqa: BLE001 -- hloc (h5py, pycolmap, ...) absent: same protocol, locally

    class BaseModel(nn.Module, metaclass=ABCMeta):
        default_conf = {}
        required_inputs = []

        def __init__(self, conf):
            super().__init__()
            self.conf = conf = {**self.default_conf, **conf}
            self.required_inputs = copy(self.required_inputs)
            self._init(conf)
            sys.stdout.flush()

        def forward(self, data):
            for key in self.required_inputs:
                assert key in data, "Missing key {} in data".format(key)
            return self._forward(data)

        @abstractmethod
        def _init(self, conf):
            raise NotImplementedError

        @abstractmethod
        def _forward(self, data):
            raise NotImplementedError


def dynamic_load(root, model):
    """hloc/utils/base_model.py:36-47, for exercising the plugin lookup without hloc"""
    import inspect
    module_path = f"{root.__name__}.{model}"
    module = __import__(module_path, fromlist=[""])
    classes = inspect.getmembers(module, inspect.isclass)
    classes = [c for c in classes if c[1].__module__ == module_path]
    classes = [c for c in classes if issubclass(c[1], BaseModel)]
    assert len(classes) == 1, classes
    return classes[0][1]
