"""`GPNeRFModel` -- inference-side mirror of reference gpnerf/model.py:20-209 (SURVEY.md section 8(f)
rank 4): the container the drivers hand to `render_rays` / `render_single_image` / `log_view`, holding
`.net_coarse`, `.net_fine`, `.feature_net`, `.feature_fpn`, `.sem_seg_head`, and the checkpoint
format of the reference (`torch.save` of {optimizer, scheduler, net_coarse, feature_net, feature_fpn,
sem_seg_head[, net_fine]}, model.py:133-146) so that a reference `.pth` loads unchanged.

Out of scope here (training container): optimizer, scheduler, DDP wrapping, SyncBatchNorm conversion
(a no-op in the reference too: its nets use Instance/Layer/GroupNorm).  Checkpoints are read with
`torch.load(weights_only=True)`: nothing from the file is executed.
"""
import os

import torch

from .feature_network import FPN, ResUNet
from .semantic_branch import NeRFSemSegFPNHead
from .transformer_network import GNT


class GPNeRFModel(object):
    def __init__(self, args, load_opt=False, load_scheduler=False):
        self.args = args
        device = torch.device("cuda:{}".format(getattr(args, "local_rank", 0)))
        n_imp = getattr(args, "N_importance", 0)
        self.net_coarse = GNT(args, in_feat_ch=args.coarse_feat_dim, posenc_dim=3 + 3 * 2 * 10,
                              viewenc_dim=3 + 3 * 2 * 10, ret_alpha=n_imp > 0).to(device)
        if getattr(args, "single_net", True):
            self.net_fine = None
        else:
            self.net_fine = GNT(args, in_feat_ch=args.fine_feat_dim, posenc_dim=3 + 3 * 2 * 10,
                                viewenc_dim=3 + 3 * 2 * 10, ret_alpha=True).to(device)
        self.feature_net = ResUNet(coarse_out_ch=args.coarse_feat_dim,
                                   fine_out_ch=getattr(args, "fine_feat_dim", args.coarse_feat_dim),
                                   single_net=getattr(args, "single_net", True)).to(device)
        self.feature_fpn = FPN(in_channels=[64, 64, 128, 256], out_channels=128, concat_out=True).to(device)
        self.sem_seg_head = NeRFSemSegFPNHead(args).to(device)
        self.optimizer = self.scheduler = None  # training state: not part of the rendering path
        self.device = device
        self.switch_to_eval()
        out_folder = os.path.join(getattr(args, "rootdir", "."), "out", getattr(args, "expname", ""))
        self.start_step = self.load_from_ckpt(out_folder, load_opt=load_opt, load_scheduler=load_scheduler)

    def _nets(self):
        nets = {"net_coarse": self.net_coarse, "feature_net": self.feature_net,
                "feature_fpn": self.feature_fpn, "sem_seg_head": self.sem_seg_head}
        if self.net_fine is not None:
            nets["net_fine"] = self.net_fine
        return nets

    def switch_to_eval(self):
        for n in self._nets().values():
            n.eval()

    def switch_to_train(self):
        """The reference flips Dropout back on here (model.py:125-131); this package is inference only
        (GNT.forward has eval semantics regardless), so the call only keeps driver code working."""

    def save_model(self, filename):
        """reference model.py:133-146 (optimizer / scheduler entries are empty: nothing is trained here)."""
        to_save = {"optimizer": {}, "scheduler": {}}
        to_save.update({k: n.state_dict() for k, n in self._nets().items()})
        torch.save(to_save, filename)

    def load_model(self, filename, load_opt=False, load_scheduler=False):
        """reference model.py:148-173: net_coarse non-strict, the other nets strict; optimizer and
        scheduler states are skipped."""
        if load_opt or load_scheduler:
            raise NotImplementedError("optimizer / scheduler state belongs to the training container "
                                      "(out of scope); load with load_opt=False, load_scheduler=False")
        to_load = torch.load(filename, map_location=self.device, weights_only=True)
        self.net_coarse.load_state_dict(to_load["net_coarse"], strict=False)
        for k in ("feature_net", "feature_fpn", "sem_seg_head", "net_fine"):
            net = getattr(self, k)
            if net is not None and k in to_load.keys():
                net.load_state_dict(to_load[k], strict=True)

    def load_from_ckpt(self, out_folder, load_opt=False, load_scheduler=False, force_latest_ckpt=False):
        """reference model.py:175-209: `args.ckpt_path` if it is a file, else the lexicographically last
        `*.pth` of `out_folder`; returns the step parsed from the file name (chars [-10:-4])."""
        ckpts = []
        if os.path.exists(out_folder):
            ckpts = [os.path.join(out_folder, f) for f in sorted(os.listdir(out_folder)) if f.endswith(".pth")]
        ckpt_path = getattr(self.args, "ckpt_path", None)
        if ckpt_path is not None and not force_latest_ckpt:
            if os.path.isfile(ckpt_path):
                ckpts = [ckpt_path]
        step = 0
        if len(ckpts) > 0 and not getattr(self.args, "no_reload", False):
            fpath = ckpts[-1]
            self.load_model(fpath, load_opt, load_scheduler)
            try:
                step = int(fpath[-10:-4])
            except ValueError:
                step = 0
            print("Reloading from {}, starting at step={}".format(fpath, step))
        return step
