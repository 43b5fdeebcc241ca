from .dkm import DKMv3, RegressionMatcher, gim_dkm_inference  # noqa: F401
