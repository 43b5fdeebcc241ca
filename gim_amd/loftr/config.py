"""gim_loftr configuration with the reference's surface (`networks/loftr/config.py:3-77`,
`networks/loftr/misc.py:13-16`) but without the yacs dependency: `get_cfg_defaults()` returns a
CfgNode-like object with UPPER-case attribute access and `.clone()`, `lower_config()` turns it into the
lower-case dict that `LoFTR(config)` takes (`demo.py:333-335`: LoFTR(lower_config(get_cfg_defaults())['loftr']))."""


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return CfgNode({k: (v.clone() if isinstance(v, CfgNode) else v) for k, v in self.items()})

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, CfgNode) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_other_cfg(v)
            else:
                self[k] = v


def _defaults():
    CN = CfgNode
    c = CN()
    c.TEMP_BUG_FIX = True
    c.LOFTR = CN()
    c.LOFTR.WEIGHT = None
    c.LOFTR.BACKBONE_TYPE = "ResNetFPN"
    c.LOFTR.RESOLUTION = (8, 2)
    c.LOFTR.FINE_WINDOW_SIZE = 5
    c.LOFTR.FINE_CONCAT_COARSE_FEAT = False
    c.LOFTR.RESNETFPN = CN(INITIAL_DIM=128, BLOCK_DIMS=[64, 128, 196, 256, 512, 1024])
    c.LOFTR.COARSE = CN(D_MODEL=256, NHEAD=8, LAYER_NAMES=4, ATTENTION="linear")
    c.LOFTR.MATCH_COARSE = CN(THR=0.2, BORDER_RM=2, MATCH_TYPE="dual_softmax", DSMAX_TEMPERATURE=0.1,
                              SKH_ITERS=3, SKH_INIT_BIN_SCORE=1.0, SKH_PREFILTER=False,
                              TRAIN_COARSE_PERCENT=0.2, TRAIN_PAD_NUM_GT_MIN=200, SPARSE_SPVS=False)
    c.LOFTR.FINE = CN(D_MODEL=128, NHEAD=8, LAYER_NAMES=1, ATTENTION="linear")
    c.LOFTR.LOSS = CN(COARSE_TYPE="focal", COARSE_WEIGHT=1.0, FOCAL_ALPHA=0.25, FOCAL_GAMMA=2.0,
                      POS_WEIGHT=1.0, NEG_WEIGHT=1.0, FINE_TYPE="l2_with_std", FINE_WEIGHT=1.0,
                      FINE_CORRECT_THR=1.0, OVERLAP_WEIGHT=20.0, OVERLAP_FOCAL_ALPHA=0.25,
                      OVERLAP_FOCAL_GAMMA=5.0)
    return c


def get_cfg_defaults():
    return _defaults()


def lower_config(cfg):
    if not isinstance(cfg, CfgNode):
        return cfg
    return {k.lower(): lower_config(v) for k, v in cfg.items()}
