"""Checkpoint files in the reference's format (common/utils/model_io.py:9-82): a torch.save'd dict
{'epoch', 'state_dict', 'optimizer', 'loss'} named `<dir>/model_epochNNN.pt.tar` or `<dir>/<name>.pt.tar`, parameter names
without the DistributedDataParallel `module.` prefix.  The mirror models keep the reference's parameter and buffer names, so a
checkpoint trained with the reference loads here and the other way round."""
import os.path as osp

import torch

from .cfgs_utils import valid_key_in_cfgs


def _log(logger, msg, level='info'):
    if logger is not None:
        logger.add_log(msg, level=level) if level != 'info' else logger.add_log(msg)


def _world_size(cfgs):
    return int(cfgs.dist.world_size) if (cfgs is not None and valid_key_in_cfgs(cfgs, 'dist')) else 1


def load_model(logger, model, optimizer, path, cfgs, strict=False):
    """Load `path` into `model` (+ optimizer and epoch when cfgs.progress.start_epoch < 0, i.e. resume).  Entries whose shape
    differs from the model's are skipped with a warning (partial initialisation); a DDP-wrapped model gets the `module.` prefix."""
    ckpt = torch.load(path, map_location='cpu')
    have = model.state_dict()
    prefix = 'module.' if _world_size(cfgs) > 1 else ''
    state = {}
    for name, value in ckpt['state_dict'].items():
        key = prefix + name
        if key in have and tuple(have[key].shape) != tuple(value.shape):
            _log(logger, 'key {} skipped because of size mismatch.'.format(name), level='warning')
            continue
        state[key] = value
    model.load_state_dict(state, strict=strict)
    resume = (cfgs is not None and valid_key_in_cfgs(cfgs, 'progress') and valid_key_in_cfgs(cfgs.progress, 'start_epoch')
              and cfgs.progress.start_epoch < 0)
    if resume:
        if optimizer is not None and 'optimizer' in ckpt:
            optimizer.load_state_dict(ckpt['optimizer'])
        cfgs.progress.start_epoch = max(0, ckpt['epoch'])
    _log(logger, 'Successfully loaded checkpoint from {} (at epoch {})... Keep Train: {}'.format(path, ckpt.get('epoch'), resume))
    return model


def save_model(logger, model, optimizer, epoch, loss, model_dir, cfgs, spec_name=None):
    """-> path of the written `*.pt.tar`"""
    net = model.module if _world_size(cfgs) > 1 else model
    name = '{}.pt.tar'.format(spec_name) if spec_name is not None else 'model_epoch{:03d}.pt.tar'.format(epoch)
    path = osp.join(model_dir, name)
    torch.save({'epoch': epoch, 'state_dict': net.state_dict(), 'optimizer': optimizer.state_dict(), 'loss': loss}, path)
    _log(logger, 'Saved model at {} ...'.format(path))
    return path
