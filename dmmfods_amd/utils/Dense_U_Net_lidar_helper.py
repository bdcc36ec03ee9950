"""Config + metric helpers with the reference's names and field layout
(reference: dmmfods/utils/Dense_U_Net_lidar_helper.py:24-228 config/json, :311-401 metrics).

Only what the hot path needs is here: the nested config object every constructor takes, and the metric
definitions.  The Waymo tfrecord conversion and ground-truth rendering of the reference are out of scope."""
import json
import os
from datetime import datetime
from os.path import isfile, join

import torch


class AttrDict(dict):
    """Recursive attribute dict (stands in for easydict.EasyDict, which the reference uses, H:9)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def load_json_file(filepath):
    if not isfile(filepath):
        raise FileNotFoundError(filepath)
    with open(filepath, "r") as jf:
        return json.load(jf)


def save_json_file(filepath, save_file, indent=None):
    with open(filepath, "w") as jf:
        json.dump(save_file, jf, indent=indent)
    return 1


def create_config(host_dir):
    """Same keys and defaults as the reference's create_config (H:84-211)."""
    host_dir = host_dir or "/content/drive/My Drive/Colab Notebooks/DeepCV_Packages"
    cfg = {"dir": {"hosting": host_dir}}
    cfg["scripts"] = {"model": "Dense_U_Net_lidar.py", "utils": "Dense_U_Net_lidar_helper.py",
                      "agent": "Dense_U_Net_lidar_Agent.py", "dataset": "WaymoData.py", "setup": "Setup.ipynb"}
    cfg["model"] = {"growth_rate": 32, "block_config": (6, 12, 24, 16), "num_init_features": 64,
                    "stream_1_in_channels": 3, "stream_2_in_channels": 1, "concat_before_block_num": 2,
                    "num_layers_before_blocks": 4, "bn_size": 4, "drop_rate": 0, "num_classes": 3,
                    "memory_efficient": False}
    cfg["loss"] = {"alpha": 1, "gamma": 2, "logits": True, "reduce": False, "skip_v_every_n_its": False,
                   "skip_p_every_n_its": False, "skip_b_every_n_its": False}
    cfg["loader"] = {"mode": "train", "batch_size": None, "pin_memory": True, "num_workers": 4,
                     "async_loading": True, "drop_last": False}
    cfg["optimizer"] = {"type": "Adam", "learning_rate": 1e-3, "beta1": 0.9, "beta2": 0.999, "eps": 1e-08,
                        "amsgrad": False, "weight_decay": 0,
                        "lr_scheduler": {"want": False, "every_n_epochs": 30, "gamma": 0.1}}
    cfg["dataset"] = {"batch_size": 32, "label": {"1": "TYPE_VEHICLE", "2": "TYPE_PEDESTRIAN", "4": "TYPE_CYCLIST"},
                      "images": {"original.size": (3, 1920, 1280), "size": (3, 192, 128)},
                      "datatypes": ["images", "lidar", "labels", "heat_maps"], "file_list_name": "file_list.json"}
    cfg["agent"] = {"seed": 123, "max_epoch": 100, "iou_threshold": 0.7,
                    "checkpoint": {"epoch": "epoch", "train_iteration": "train_iteration",
                                   "val_iteration": "val_iteration", "best_val_iou": "best_val_iou",
                                   "state_dict": "state_dict", "optimizer": "optimizer"},
                    "best_checkpoint_name": "best_checkpoint.pth.tar"}
    root = join(host_dir, "DMMFODS", "dmmfods")
    cfg["dir"]["root"] = root
    for sub in ("agents", "graphs", "utils", "datasets", "configs", "experiments"):
        cfg["dir"][sub] = join(root, sub)
    cfg["dir"]["graphs"] = {"models": join(cfg["dir"]["graphs"], "models")}
    cfg["dir"]["data"] = {"root": join(host_dir, "data"), "file_lists": join(root, "data")}
    run = datetime.now().strftime("%Y-%m-%d-%H-%M")
    cfg["dir"]["current_run"] = {"summary": join(cfg["dir"]["experiments"], run, "summary"),
                                 "checkpoints": join(cfg["dir"]["experiments"], run, "checkpoints")}
    return cfg


def load_config(loading_dir, file_name):
    path = join(loading_dir, file_name)
    return load_json_file(path) if isfile(path) else None


def get_config(host_dir="", file_name="config.json"):
    cfg = load_config(join(host_dir, "DMMFODS", "dmmfods", "configs"), file_name)
    if cfg is None:
        cfg = create_config(host_dir)
    return AttrDict(cfg)


def save_config(config, file_name="config.json"):
    os.makedirs(config.dir.configs, exist_ok=True)
    save_json_file(join(config.dir.configs, file_name), config, indent=4)


def set_current_run(config, current_run):
    for key in ("summary", "checkpoints"):
        parts = config.dir.current_run[key].split("/")[:-2]
        config.dir.current_run[key] = "/" + os.path.join(*parts, current_run, key)
    return config


# ---------------------------------------------------------------- metrics (H:311-401)
def iou_from_counts(inter, union):
    """inter, union: (B, C) counts -> IoU with 0/0 = NaN, as intersection/union of float tensors gives."""
    return inter.float() / union.float()


def compute_IoU_whole_img_per_class(ground_truth_map, estimated_heat_map, threshold):
    est, gt = estimated_heat_map >= threshold, ground_truth_map >= threshold
    inter = torch.sum(est & gt, dim=(1, 2)).float()
    union = torch.sum(est | gt, dim=(1, 2)).float()
    return inter / union


def compute_IoU_whole_img_batch(ground_truth_map_batch, estimated_heat_map_batch, threshold=0.7):
    """(B, C) whole-image IoU; unlike the reference (H:359-363) the result stays on the inputs' device and
    is produced by one reduction, so it does not force a host sync per sample."""
    est, gt = estimated_heat_map_batch >= threshold, ground_truth_map_batch >= threshold
    inter = torch.sum(est & gt, dim=(2, 3)).float()
    union = torch.sum(est | gt, dim=(2, 3)).float()
    return inter / union


def compute_accuracy(ground_truth, prediction, threshold=0.7):
    if ground_truth.dim() == 3:
        axes, ncls = (1, 2), ground_truth.shape[0]
    elif ground_truth.dim() == 4:
        axes, ncls = (0, 2, 3), ground_truth.shape[1]
    else:
        raise ValueError("Number of dimensions must be either 3 or 4, you gave " + str(ground_truth.dim()))
    eq = (prediction >= threshold) == (ground_truth >= threshold)
    return torch.sum(eq, dim=axes) / (ground_truth.numel() / ncls)
