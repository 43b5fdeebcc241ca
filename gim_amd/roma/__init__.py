from .roma import RegressionMatcher, RoMa, gim_roma_inference, random_dinov2_weights  # noqa: F401
