"""`from config import cfg` surface of the reference (config/__init__.py, config/defaults.py).

yacs is not available offline, so `CfgNode` below is a minimal attribute-dict with the four
methods train_net.py uses (merge_from_file / merge_from_list / freeze / clone, train_net.py:46-50).
Only the keys that the hot path or its callers read are declared; values are the reference's
defaults (config/defaults.py:1-217).
"""
from __future__ import annotations

import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {k} on a frozen CfgNode")
        self[k] = v

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        n = CfgNode()
        for k, v in self.items():
            dict.__setitem__(n, k, copy.deepcopy(v, memo))
        return n

    def _merge(self, other: dict, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError(f"Non-existent config key: {path}{k}")
            if isinstance(self[k], CfgNode):
                self[k]._merge(v, path + k + ".")
            else:
                self[k] = _coerce(v, self[k], path + k)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {key}")
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = _coerce(val, node[parts[-1]], key)


def _coerce(v, old, key):
    if isinstance(old, tuple) and isinstance(v, list):
        return tuple(v)
    if isinstance(old, list) and isinstance(v, tuple):
        return list(v)
    if isinstance(old, float) and isinstance(v, int) and not isinstance(v, bool):
        return float(v)
    if isinstance(old, str) and isinstance(v, str) and v.startswith("(") and v.endswith(")"):
        return v.strip("()'\" ")          # yaml "('RGBNT201')" -> 'RGBNT201'
    return v


def get_defaults() -> CfgNode:
    return CfgNode({
        "MODEL": {
            "ARCH": "DeMo", "DEVICE": "cuda", "DEVICE_ID": "0", "NAME": "DeMo",
            "PRETRAIN_PATH_T": "", "NECK": "bnneck", "IF_WITH_CENTER": "no", "ID_LOSS_TYPE": "softmax",
            "ID_LOSS_WEIGHT": 1.0, "TRIPLET_LOSS_WEIGHT": 1.0, "METRIC_LOSS_TYPE": "triplet", "DIST_TRAIN": False,
            "PROMPT": False, "ADAPTER": False, "FROZEN": False, "HDM": False, "ATM": False,
            "USE_SACR": False, "USE_SDTPS": False, "USE_LIF": False, "USE_DGAF": False, "USE_MULTIMODAL_SACR": False,
            "USE_FRCA": None, "DGAF_VERSION": "v3", "DGAF_TAU": 1.0, "DGAF_INIT_ALPHA": 0.5, "DGAF_NUM_HEADS": 8,
            "IF_LABELSMOOTH": "on", "DIRECT": 1, "DROP_PATH": 0.1, "DROP_OUT": 0.0, "ATT_DROP_RATE": 0.0,
            "TRANSFORMER_TYPE": "vit_base_patch16_224", "STRIDE_SIZE": [16, 16], "GLOBAL_LOCAL": False, "HEAD": 12,
            "SIE_COE": 3.0, "SIE_CAMERA": True, "SIE_VIEW": False, "NO_MARGIN": True,
        },
        "INPUT": {"SIZE_TRAIN": [256, 128], "SIZE_TEST": [256, 128], "PROB": 0.5, "RE_PROB": 0.5,
                  "PIXEL_MEAN": [0.5, 0.5, 0.5], "PIXEL_STD": [0.5, 0.5, 0.5], "PADDING": 10},
        "DATASETS": {"NAMES": "RGBNT201", "ROOT_DIR": "./data"},
        "DATALOADER": {"NUM_WORKERS": 14, "SAMPLER": "softmax_triplet", "NUM_INSTANCE": 16},
        "SOLVER": {"OPTIMIZER_NAME": "SGD", "MAX_EPOCHS": 120, "BASE_LR": 0.009, "LARGE_FC_LR": False, "MOMENTUM": 0.9,
                   "MARGIN": 0.3, "CENTER_LR": 0.5, "CENTER_LOSS_WEIGHT": 0.0005, "WEIGHT_DECAY": 0.0001,
                   "WEIGHT_DECAY_BIAS": 0.0001, "GAMMA": 0.1, "STEPS": (40, 70), "WARMUP_FACTOR": 0.01, "WARMUP_ITERS": 10,
                   "WARMUP_METHOD": "linear", "LR_SCHEDULER": "cosine", "SEED": 1234, "CHECKPOINT_PERIOD": 10,
                   "LOG_PERIOD": 10, "EVAL_PERIOD": 1, "IMS_PER_BATCH": 128},
        "TEST": {"IMS_PER_BATCH": 256, "RE_RANKING": "no", "WEIGHT": "", "NECK_FEAT": "before", "FEAT_NORM": "yes",
                 "MISS": "None"},
        "OUTPUT_DIR": "./test",
    })


cfg = get_defaults()
