from .roma import RegressionMatcher, RoMa, gim_roma_inference  # noqa: F401
