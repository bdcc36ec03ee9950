"""Training / validation driver with the reference agent's surface (reference: dmmfods/agents/Dense_U_Net_lidar_Agent.py:21-450):
``run / train / train_one_epoch / validate / save_checkpoint / load_checkpoint / finalize``, the checkpoint dict keyed by
``config.agent.checkpoint.*``, optional StepLR, best-checkpoint selection by mean validation IoU.

Differences, all on purpose: the step body uses the fused HIP tail (``model.loss_backward``), so loss sums, IoU and accuracy
come back as device tensors without the per-iteration host syncs of A:252-260 / H:359-363; TensorBoard is optional (not
installed here); ``compute_dtype`` / data parallelism are new."""
import logging
import os
import warnings
from datetime import datetime
from pathlib import Path

import torch

from ..datasets.WaymoData import WaymoDataset_Loader
from ..graphs.models.Dense_U_Net_lidar import densenet121_u_lidar
from ..optim import FusedAdam
from ..utils import Dense_U_Net_lidar_helper as utils

try:  # pragma: no cover - tensorboard is not installed in the build image
    from torch.utils.tensorboard import SummaryWriter
except Exception:  # noqa: BLE001
    SummaryWriter = None

CLASS_NAMES = ("Vehicle", "Pedestrian", "Cyclist")


class _NullWriter:
    def add_scalars(self, *a, **k):
        pass

    add_hparams = add_scalars

    def close(self):
        pass


class _StepLR:
    """lr = lr0 * gamma ** (epoch // step_size), stepped once per epoch (torch.optim.lr_scheduler.StepLR semantics)."""

    def __init__(self, optimizer, step_size, gamma):
        self.opt, self.step_size, self.gamma = optimizer, step_size, gamma
        self.base = [g["lr"] for g in optimizer.param_groups]
        self.epoch = 0

    def step(self):
        self.epoch += 1
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = b * self.gamma ** (self.epoch // self.step_size)


class Dense_U_Net_lidar_Agent:
    def __init__(self, config=None, torchvision_init=True, compute_dtype=None, data_loader=None, loss=None):
        self.logger = logging.getLogger("Agent")
        # the reference always builds DenseNet-121 (A:44); torchvision_init=True would download ImageNet weights (no network)
        self.model = densenet121_u_lidar(pretrained=False, config=config, compute_dtype=compute_dtype)
        self.config = self.model.config
        self.data_loader = data_loader if data_loader is not None else WaymoDataset_Loader(self.config)
        # A:54.  The step uses the fused tail; a FocalLoss / ClassWiseFocalLoss (reference L:9-91) becomes its loss epilogue.
        self.loss = loss if loss is not None else torch.nn.BCEWithLogitsLoss(reduction="none")
        if hasattr(self.loss, "attach"):
            self.loss.attach(self.model)
        elif not isinstance(self.loss, torch.nn.BCEWithLogitsLoss):
            raise ValueError("loss must be BCEWithLogitsLoss(reduction='none') or a dmmfods_amd focal loss")
        o = self.config.optimizer
        self.optimizer = FusedAdam(self.model, lr=o.learning_rate, betas=(o.beta1, o.beta2), eps=o.eps,
                                   weight_decay=o.weight_decay, amsgrad=o.amsgrad)
        self.lr_scheduler = None
        if o.lr_scheduler.want:
            self.lr_scheduler = _StepLR(self.optimizer, o.lr_scheduler.every_n_epochs, o.lr_scheduler.gamma)
        self.current_epoch = self.current_train_iteration = self.current_val_iteration = 0
        self.best_val_iou = 0
        # per-epoch averages as the reference logs them (A:301-307, A:392-398), kept for callers / tests: one dict per epoch
        self.train_history, self.val_history = [], []
        self.cuda = torch.cuda.is_available()
        if not self.cuda:
            raise RuntimeError("dmmfods_amd computes on the GPU only (no CPU fallback)")
        self.device = torch.device("cuda")
        torch.cuda.manual_seed_all(self.config.agent.seed)
        self.model = self.model.to(self.device)
        if not torchvision_init:
            self.load_checkpoint()
        Path(self.config.dir.current_run.summary).mkdir(exist_ok=True, parents=True)
        mk = (lambda: SummaryWriter(log_dir=self.config.dir.current_run.summary, comment="Dense_U_Net")) if SummaryWriter else _NullWriter
        self.train_summary_writer, self.val_summary_writer = mk(), mk()

    # ------------------------------------------------------------------ checkpoints (A:96-163)
    def save_checkpoint(self, filename="checkpoint.pth.tar", is_best=False):
        k = self.config.agent.checkpoint
        state = {k.epoch: self.current_epoch, k.train_iteration: self.current_train_iteration,
                 k.val_iteration: self.current_val_iteration, k.best_val_iou: self.best_val_iou,
                 k.state_dict: self.model.state_dict(), k.optimizer: self.optimizer.state_dict()}
        if is_best:
            filename = self.config.agent.best_checkpoint_name
        Path(self.config.dir.current_run.checkpoints).mkdir(exist_ok=True, parents=True)
        torch.save(state, os.path.join(self.config.dir.current_run.checkpoints, filename))

    def load_checkpoint(self, filename=None):
        filename = filename or self.config.agent.best_checkpoint_name
        path = os.path.join(self.config.dir.current_run.checkpoints, filename)
        k = self.config.agent.checkpoint
        try:
            ck = torch.load(path, map_location="cpu")
        except OSError:
            warnings.warn("No checkpoint exists from {}. Skipping...".format(path))
            self.logger.info("**First time to train**")
            return
        self.current_epoch = ck[k.epoch]
        self.current_train_iteration = ck[k.train_iteration]
        self.current_val_iteration = ck[k.val_iteration]
        self.best_val_iou = ck[k.best_val_iou]
        self.model.load_state_dict(ck[k.state_dict])
        self.optimizer.load_state_dict(ck[k.optimizer])

    # ------------------------------------------------------------------ driver (A:165-213)
    def run(self):
        print("starting " + self.config.loader.mode + " at " + str(datetime.now()))
        try:
            if self.config.loader.mode == "test":
                with torch.no_grad():
                    self.validate()
            else:
                self.train()
        except KeyboardInterrupt:
            self.logger.info("You have entered CTRL+C.. Wait to finalize")

    def train(self):
        self.config.loss.func = str(self.loss)
        self.config.optimizer.func = "FusedAdam(" + str(self.optimizer.defaults) + ")"
        self.save_hparams_json()
        for epoch in range(self.current_epoch, self.config.agent.max_epoch):
            self.current_epoch = epoch
            self.train_one_epoch()
            with torch.no_grad():
                avg_val_iou_per_class = self.validate()
            val_iou = sum(avg_val_iou_per_class) / len(avg_val_iou_per_class)
            is_best = val_iou > self.best_val_iou
            if is_best:
                self.best_val_iou = val_iou
            self.save_checkpoint(is_best=is_best)
        self.train_summary_writer.close()
        self.val_summary_writer.close()

    def _to_device(self, *tensors):
        nb = bool(self.config.loader.async_loading)
        return tuple(t.to(self.device, non_blocking=nb) for t in tensors)

    @staticmethod
    def _batch_metrics(m):
        """Per-class IoU (NaN-mean over samples, NaN -> 0), NaN counts, accuracy: A:252-260 without leaving the device."""
        iou = m["iou_per_instance_per_class"]
        nan = torch.isnan(iou)
        cnt = (~nan).sum(dim=0).clamp_min(1)
        iou_pc = torch.where(nan, torch.zeros_like(iou), iou).sum(dim=0) / cnt
        iou_pc = torch.where((~nan).any(dim=0), iou_pc, torch.zeros_like(iou_pc))
        return iou_pc, nan.sum(dim=0), m["acc_per_class"]

    def _log(self, writer, tag, loss_pc, acc_pc, iou_pc, it):
        for name, vals in (("Loss", loss_pc), ("Accuracy", acc_pc), ("IoU", iou_pc)):
            d = {c: vals[i] for i, c in enumerate(CLASS_NAMES[: len(vals)])}
            d["Overall"] = torch.mean(vals)
            writer.add_scalars(f"{tag}/{name}", d, it)

    def train_one_epoch(self):
        self.model.train()
        n = self.data_loader.train_iterations
        nc = self.config.model.num_classes
        ep = {k: torch.zeros((n, nc), device=self.device) for k in ("loss", "iou", "nans", "acc")}
        for b, (image, lidar, ht_map) in enumerate(self.data_loader.train_loader):
            image, lidar, ht_map = self._to_device(image, lidar, ht_map)
            with torch.no_grad():
                self.model(image, lidar)                      # A:244
            m = self.model.loss_backward(ht_map)              # A:247-264
            self.optimizer.step()                             # A:265
            iou_pc, nans, acc_pc = self._batch_metrics(m)
            ep["loss"][b], ep["iou"][b], ep["nans"][b], ep["acc"][b] = m["loss_per_class"], iou_pc, nans, acc_pc
            self._log(self.train_summary_writer, "Training", m["loss_per_class"], acc_pc, iou_pc, self.current_train_iteration)
            self.current_train_iteration += 1
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        self.train_history.append({"epoch": self.current_epoch, "loss": ep["loss"].mean(0).cpu(), "iou": ep["iou"].mean(0).cpu(),
                                   "nans": ep["nans"].sum(0).cpu(), "acc": ep["acc"].mean(0).cpu()})
        self.logger.info("Training at Epoch-%d | Average Loss: %s | Average IoU: %s | Number of NaNs: %s | Average Accuracy: %s",
                         self.current_epoch, ep["loss"].mean(0).tolist(), ep["iou"].mean(0).tolist(), ep["nans"].sum(0).tolist(),
                         ep["acc"].mean(0).tolist())

    def validate(self):
        self.model.eval()
        n = self.data_loader.valid_iterations
        nc = self.config.model.num_classes
        ep = {k: torch.zeros((n, nc), device=self.device) for k in ("loss", "iou", "nans", "acc")}
        with torch.no_grad():
            for b, (image, lidar, ht_map) in enumerate(self.data_loader.valid_loader):
                image, lidar, ht_map = self._to_device(image, lidar, ht_map)
                prediction = self.model(image, lidar)
                m = self.model.loss_metrics(prediction, ht_map)
                iou_pc, nans, acc_pc = self._batch_metrics(m)
                ep["loss"][b], ep["iou"][b], ep["nans"][b], ep["acc"][b] = m["loss_per_class"], iou_pc, nans, acc_pc
                self._log(self.val_summary_writer, "Validation", m["loss_per_class"], acc_pc, iou_pc, self.current_val_iteration)
                self.current_val_iteration += 1
        avg_iou = ep["iou"].mean(0).tolist()
        self.val_history.append({"epoch": self.current_epoch, "loss": ep["loss"].mean(0).cpu(), "iou": ep["iou"].mean(0).cpu(),
                                 "nans": ep["nans"].sum(0).cpu(), "acc": ep["acc"].mean(0).cpu()})
        self.logger.info("Validation at Epoch-%d | Average Loss: %s | Average IoU: %s | Number of NaNs: %s | Average Accuracy: %s",
                         self.current_epoch, ep["loss"].mean(0).tolist(), avg_iou, ep["nans"].sum(0).tolist(), ep["acc"].mean(0).tolist())
        return avg_iou

    def save_hparams_json(self):
        hp = {"loss": dict(self.config.loss), "optimizer": {k: v for k, v in dict(self.config.optimizer).items()}}
        Path(self.config.dir.current_run.summary).mkdir(exist_ok=True, parents=True)
        utils.save_json_file(os.path.join(self.config.dir.current_run.summary, "hyperparams.json"), hp, indent=4)

    def finalize(self):
        self.logger.info("Please wait while finalizing the operation.. Thank you")
        self.train_summary_writer.close()
        self.val_summary_writer.close()
        print("ending " + self.config.loader.mode + " at " + str(datetime.now()))
