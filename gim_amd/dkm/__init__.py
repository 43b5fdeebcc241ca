from .dkm import DKMv3, RegressionMatcher  # noqa: F401
