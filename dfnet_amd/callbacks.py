"""Host-side mirror of /root/reference/script/dm/callbacks.py: EarlyStopping on the validation loss (or PSNR), saving
`checkpoint.pt` / `checkpoint-<epoch>-<loss>.pt` under <basedir>/<model_name>/ whenever the monitored value improves."""
import os

import numpy as np
import torch


class EarlyStopping:
    def __init__(self, args, patience=50, verbose=False, delta=0):
        self.val_on_psnr = args.val_on_psnr
        self.patience, self.verbose, self.delta = patience, verbose, delta
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = np.inf
        self.out_folder = os.path.join(args.basedir, args.model_name)
        self.ckpt_save_path = os.path.join(self.out_folder, 'checkpoint.pt')
        os.makedirs(self.out_folder, exist_ok=True)

    def __call__(self, val_loss, model, epoch=-1, save_multiple=False, save_all=False, val_psnr=None):
        value = val_psnr if self.val_on_psnr else val_loss
        score = val_psnr if self.val_on_psnr else -val_loss
        if self.best_score is None or score >= self.best_score + self.delta:
            self.best_score = score
            self.save_checkpoint(value, model, epoch=epoch, save_multiple=save_multiple)
            self.counter = 0
            return
        self.counter += 1
        if self.counter >= self.patience:
            self.early_stop = True
        if save_all:
            self.save_checkpoint(value, model, epoch=epoch, save_multiple=True, update_best=False)

    def save_checkpoint(self, val_loss, model, epoch=-1, save_multiple=False, update_best=True):
        if self.verbose:
            print(f'Validation loss decreased ({self.val_loss_min:.6f} --> {val_loss:.6f}).  Saving model ...')
        path = self.ckpt_save_path
        if save_multiple:
            path = path[:-3] + f'-{epoch:04d}-{val_loss:.4f}.pt'
        from .dist import rank_world, sync_buffers
        sync_buffers(model)        # data-parallel training: parameters are identical, BatchNorm's running statistics are per rank
        if rank_world()[0] == 0:   # ... rank 0 writes
            torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, path)
        if update_best:
            self.val_loss_min = val_loss

    def isBestModel(self):
        return self.counter == 0
