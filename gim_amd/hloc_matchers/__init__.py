"""Drop-in `hloc.matchers` plugins backed by the HIP engine.

The reference's hloc looks a matcher up with `dynamic_load(hloc.matchers, conf['model']['name'])`
(`hloc/utils/base_model.py:36-47`): it imports `hloc.matchers.<name>` and takes the unique `BaseModel` subclass defined in
that module.  Copy (or symlink) `gim_dkm_hip.py` into the reference's `hloc/matchers/` -- or put this package on the path as
`hloc.matchers` -- and select it with `matcher_conf['model']['name'] = 'gim_dkm_hip'`; nothing else in `hloc/match_dense.py`
changes.  hloc's own `BaseModel` is the base class (so `issubclass` holds inside dynamic_load); without hloc on the path the plugin
modules do not import (the tests provide a protocol stand-in under tests/hloc_stub/).
"""
