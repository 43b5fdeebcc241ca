"""Base class of the hloc plugins in this package: hloc's OWN `hloc.utils.base_model.BaseModel` (hloc/utils/base_model.py:8-33),
so that `hloc.utils.base_model.dynamic_load` (:36-47) finds the plugin with `issubclass`.  The plugins are meant to live inside
an hloc installation; without hloc on the path importing them fails loudly here -- the package carries no
stand-in of hloc's class (the tests bring their own under tests/hloc_stub/)."""
try:
    from hloc.utils.base_model import BaseModel  # noqa: F401
except Exception as e:  # noqa: BLE001
    raise ImportError("gim_amd.hloc_matchers needs the reference's hloc package on sys.path (hloc.utils.base_model.BaseModel): "
                      f"{type(e).__name__}: {e}") from e
