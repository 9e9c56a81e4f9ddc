"""Detector settings with the field names of the reference's global ``config``
(code/detection/symbol/config.py:4-27).  The reference uses an EasyDict that is
mutated in place before the graph is built; this is the same kind of object
(attribute and item access) without the dependency."""
import numpy as np


class AttrDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def default_config():
    cfg = AttrDict()
    cfg.EPS = 1e-14
    cfg.PIXEL_MEANS = np.array([[[123.68, 116.779, 103.939]]])
    cfg.SCALES = (600,)
    cfg.MAX_SIZE = 1000
    cfg.USE_GPU_NMS = True
    cfg.GPU_ID = 0
    cfg.END2END = 1
    cfg.TEST = AttrDict()
    cfg.TEST.HAS_RPN = True
    cfg.TEST.BATCH_IMAGES = 1
    cfg.TEST.NMS = 0.3
    cfg.TEST.DEDUP_BOXES = 1.0 / 16.0
    cfg.TEST.RPN_NMS_THRESH = 0.7
    cfg.TEST.RPN_PRE_NMS_TOP_N = 1000
    cfg.TEST.RPN_POST_NMS_TOP_N = 10
    cfg.TEST.RPN_MIN_SIZE = 16
    return cfg


config = default_config()
