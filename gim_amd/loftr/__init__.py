from .config import get_cfg_defaults, lower_config  # noqa: F401
from .loftr import LoFTR  # noqa: F401
