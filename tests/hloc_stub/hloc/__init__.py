"""TEST STAND-IN for the reference's `hloc` package (absent on the test machines: h5py / pycolmap are not installed).
Only what the plugin tests touch is provided: `hloc.utils.base_model`.  `tests/conftest.py` puts this directory on
sys.path when the real hloc cannot be imported; the product package never ships or imports it."""
