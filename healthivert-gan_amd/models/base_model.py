"""BaseModel with the reference's public surface (reference models/base_model.py:8-243): option handling, schedulers,
eval/train/test, loss/visual getters, `<epoch>_net_<name>.pth` checkpoints with the reference's state-dict keys.
"""
import os
from abc import ABC, abstractmethod
from collections import OrderedDict

import torch

from . import networks
from ._backend import ddp


class BaseModel(ABC):
    def __init__(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.device = torch.device('cuda:{}'.format(self.gpu_ids[0])) if self.gpu_ids else torch.device('cpu')
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names, self.model_names, self.visual_names, self.optimizers, self.image_paths = [], [], [], [], []
        self.metric = 0

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    @abstractmethod
    def set_input(self, input):
        pass

    @abstractmethod
    def forward(self):
        pass

    @abstractmethod
    def optimize_parameters(self):
        pass

    def _nets(self):
        return [(n, getattr(self, 'net' + n)) for n in self.model_names if isinstance(n, str)]

    def setup(self, opt):
        if self.isTrain:
            self.schedulers = [networks.get_scheduler(o, opt) for o in self.optimizers]
        if not self.isTrain or opt.continue_train:
            self.load_networks('iter_%d' % opt.load_iter if opt.load_iter > 0 else opt.epoch)
        self.print_networks(opt.verbose)

    def sync_tail(self):
        """Order the current stream after work a subclass leaves pending on other streams (Pix2PixModel's data-parallel step)."""

    def eval(self):
        self.sync_tail()
        for _, net in self._nets():
            net.eval()

    def train(self):
        for _, net in self._nets():
            net.train()

    def test(self):
        with torch.no_grad():
            self.forward()
            self.compute_visuals()

    def compute_visuals(self):
        pass

    def get_image_paths(self):
        return self.image_paths

    def update_learning_rate(self):
        self.sync_tail()      # an optimiser step still queued on another stream (data-parallel schedule) must read the OLD learning rate
        old_lr = self.optimizers[0].param_groups[0]['lr']
        for s in self.schedulers:
            if self.opt.lr_policy == 'plateau':
                s.step(self.metric)
            else:
                s.step()
        print('learning rate %.7f -> %.7f' % (old_lr, self.optimizers[0].param_groups[0]['lr']))

    def get_current_visuals(self):
        self.sync_tail()
        return OrderedDict((n, getattr(self, n)) for n in self.visual_names if isinstance(n, str))

    def get_current_losses(self):
        return OrderedDict((n, float(getattr(self, 'loss_' + n))) for n in self.loss_names if isinstance(n, str))

    def save_networks(self, epoch):
        if ddp.rank() != 0:        # one process per GPU, identical weights on every rank: rank 0 writes the checkpoint
            return
        self.sync_tail()
        os.makedirs(self.save_dir, exist_ok=True)
        for name, net in self._nets():
            sd = OrderedDict((k, v.detach().cpu()) for k, v in net.state_dict().items())
            if hasattr(net.state_dict(), '_metadata'):
                sd._metadata = net.state_dict()._metadata
            torch.save(sd, os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch, name)))

    def load_networks(self, epoch):
        for name, net in self._nets():
            path = os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch, name))
            print('loading the model from %s' % path)
            sd = torch.load(path, map_location=str(self.device))
            if hasattr(sd, '_metadata'):
                del sd._metadata
            # InstanceNorm checkpoints from before torch 0.4 carried running stats the modules no longer have
            for k in list(sd.keys()):
                mod = net
                parts = k.split('.')
                for part in parts[:-1]:
                    mod = getattr(mod, part)
                if mod.__class__.__name__.startswith('InstanceNorm') and parts[-1] in ('running_mean', 'running_var', 'num_batches_tracked'):
                    if getattr(mod, parts[-1], None) is None:
                        sd.pop(k)
            net.load_state_dict(sd)

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name, net in self._nets():
            if verbose:
                print(net)
            print('[Network %s] Total number of parameters : %.3f M' % (name, sum(p.numel() for p in net.parameters()) / 1e6))
        print('-----------------------------------------------')

    def set_requires_grad(self, nets, requires_grad=False):
        for net in (nets if isinstance(nets, list) else [nets]):
            if net is not None:
                for p in net.parameters():
                    p.requires_grad = requires_grad
