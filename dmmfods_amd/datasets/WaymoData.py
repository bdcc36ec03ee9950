"""Reader for the reference's on-disk training data (reference: dmmfods/datasets/WaymoData.py:9-213, written by
utils/Dense_U_Net_lidar_helper.py:653-728): either one ``.pt`` file per sample and datatype, or "batched" files holding a
(N, 7, H, W) tensor = 3 RGB + 1 LiDAR + 3 heat-map planes.  The file-list JSON cache (``<mode>_file_list.json``) has the
reference's format, so lists crawled by the reference are reused.  Batches are handed out in pinned memory for
asynchronous H2D copies (config.loader.pin_memory / async_loading)."""
from os import listdir
from os.path import isdir, isfile, join
from pathlib import Path

import torch
from torch.utils.data import DataLoader, Dataset

from ..utils.Dense_U_Net_lidar_helper import load_json_file, save_json_file

MODES = ("train", "val", "test")


def split_batch(batch):
    """(N, 7, H, W) -> rgb (N,3,H,W), lidar (N,1,H,W), heat maps (N,3,H,W)   (reference D:87-103)."""
    return batch[:, :3], batch[:, 3:4], batch[:, 4:]


class WaymoDataset(Dataset):
    def __init__(self, mode, config):
        super().__init__()
        if mode not in MODES:
            raise ValueError("Please choose a one of the following modes: train, val, test")
        self.root = config.dir.data.root
        self.data_is_batched = config.dataset.batch_size > 1
        if config.dataset.batch_size < 1:
            raise ValueError("make sure that config.dataset.batch_size >= 1")
        cache = join(config.dir.data.file_lists, mode + "_" + config.dataset.file_list_name)
        if isfile(cache):
            self.files = load_json_file(cache)
        else:
            self.files = self._crawl_batched(mode, config) if self.data_is_batched else self._crawl_single(mode, config)
            Path(config.dir.data.file_lists).mkdir(parents=True, exist_ok=True)
            save_json_file(cache, self.files)

    def _crawl_batched(self, mode, config):
        if config.loader.batch_size is not None:
            raise ValueError("config.loader.batch_size needs to be None if loading batched dataset")
        files = []
        for sub in sorted(listdir(join(self.root, mode))):
            for name in sorted(listdir(join(self.root, mode, sub))):
                if name != "labels":
                    files.append(join(mode, sub, name))
        return files

    def _crawl_single(self, mode, config):
        files = {dt: [] for dt in config.dataset.datatypes}
        for bucket in sorted(b for b in listdir(self.root) if b.startswith("training_0")):
            for rec in listdir(join(self.root, bucket)):
                for dt in config.dataset.datatypes:
                    rel = join(bucket, rec, mode, dt)
                    if isdir(join(self.root, rel)):
                        files[dt] += [join(rel, f) for f in listdir(join(self.root, rel))]
        for i, img in enumerate(files["images"]):  # names must pair up (reference D:150-158)
            for dt in ("lidar", "heat_maps"):
                assert files[dt][i].endswith(img[-11:]), f"{i} {files[dt][i]} {img}"
        return files

    def __len__(self):
        return len(self.files) if self.data_is_batched else len(self.files["images"])

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        if self.data_is_batched:
            return split_batch(torch.load(join(self.root, self.files[idx])))
        return tuple(torch.load(join(self.root, self.files[dt][idx])) for dt in ("images", "lidar", "heat_maps"))


class WaymoDataset_Loader:
    """train_loader / valid_loader (+ *_iterations) like the reference; in 'test' mode valid_loader serves the test split."""

    def __init__(self, config):
        self.mode = config.loader.mode
        kw = dict(batch_size=config.loader.batch_size, num_workers=config.loader.num_workers,
                  pin_memory=config.loader.pin_memory, drop_last=config.loader.drop_last)

        def iters(ds):
            return len(ds) if ds.data_is_batched else (len(ds) + config.loader.batch_size) // config.loader.batch_size

        if self.mode == "train":
            train_set, valid_set = WaymoDataset("train", config), WaymoDataset("val", config)
            self.train_loader, self.valid_loader = DataLoader(train_set, **kw), DataLoader(valid_set, **kw)
            self.train_iterations, self.valid_iterations = iters(train_set), iters(valid_set)
        elif self.mode == "test":
            test_set = WaymoDataset("test", config)
            self.valid_loader = DataLoader(test_set, **kw)
            self.valid_iterations = iters(test_set)
        else:
            raise ValueError("Please choose a one of the following modes: train, val, test")
