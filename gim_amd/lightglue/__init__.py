from .superpoint import SuperPoint  # noqa: F401
from .lightglue import LightGlue  # noqa: F401
from .pipeline import gim_lightglue_inference  # noqa: F401
