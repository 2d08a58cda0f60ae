"""`s2g_body_vq` wrapper mirror (nets/smplx_body_vq.py:23-307): the legacy single-codebook VQ-VAE
(type 'bh' hard-coded at :37; non-composition form)."""
import numpy as np
import torch

from .. import train_ops, training
from .base import TrainWrapperBaseClass
from .modules import VQVAE


class TrainWrapper(TrainWrapperBaseClass):
    def __init__(self, args, config):
        self.args = args
        self.config = config
        self.device = torch.device(self.args.gpu)
        self.global_step = 0
        self.convert_to_6d = self.config.Data.pose.convert_to_6d
        self.expression = self.config.Data.pose.expression
        self.epoch = 0
        self.type = 'bh'
        self.init_params()
        self.composition = self.config.Model.composition
        in_dim = self.each_dim[1] + self.each_dim[2]
        if self.composition:
            self.g_body = VQVAE(self.each_dim[1], 64, config.Model.code_num, 1024, 2, 512)
            self.g_hand = VQVAE(self.each_dim[2], 64, config.Model.code_num, 1024, 2, 512)
        else:
            self.g = VQVAE(in_dim, 64, config.Model.code_num, 1024, 2, 512)
        self.discriminator = None
        # data_utils/lower_body.py:43-63 (c_index / c_index_6d): the 135 body+hand joints kept out of 165
        fix = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 21, 22, 23, 24, 25, 26, 30, 31, 32, 33, 34, 35]
        keep = [i for i in range(165) if i not in fix]
        self.c_index = np.asarray([v for i in keep for v in (2 * i, 2 * i + 1)] if self.convert_to_6d else keep)
        super().__init__(args, config)

    def init_optimizer(self):
        lr = self.config.Train.learning_rate.generator_learning_rate
        # optim.AdamW(betas 0.9/0.999, default weight decay 1e-2) (smplx_body_vq.py:68-86) on the HIP AdamW kernel
        mk = lambda m: train_ops.AdamW([{'params': m.parameters()}], lr=lr, betas=[0.9, 0.999], weight_decay=1e-2)
        if self.composition:
            self.g_body_optimizer = mk(self.g_body)
            self.g_hand_optimizer = mk(self.g_hand)
        else:
            self.g_optimizer = mk(self.g)

    def state_dict(self):
        if self.composition:
            return {'g_body': self.g_body.state_dict(), 'g_body_optim': self.g_body_optimizer.state_dict(),
                    'g_hand': self.g_hand.state_dict(), 'g_hand_optim': self.g_hand_optimizer.state_dict(),
                    'discriminator': None, 'discriminator_optim': None}
        return {'g': self.g.state_dict(), 'g_optim': self.g_optimizer.state_dict(), 'discriminator': None,
                'discriminator_optim': None}

    def load_state_dict(self, state_dict):
        if self.composition:
            self.g_body.load_state_dict(state_dict['g_body'])
            self.g_hand.load_state_dict(state_dict['g_hand'])
        else:
            self.g.load_state_dict(state_dict['g'])

    def __call__(self, bat):
        """One training step (smplx_body_vq.py:119-168): body+hand poses through the legacy VQ-VAE in train mode
        (batch-statistic BatchNorm, single-codebook search, commitment loss + straight-through estimator, EMA codebook
        update without re-seeding), L1 + velocity L1 + commitment loss (170-190), backward, AdamW; with
        `composition` one model and one optimizer step per part.  All arithmetic on the HIP training kernels."""
        self.global_step += 1
        poses = bat['poses'].to('cuda').to(torch.float32)
        gt_poses = poses[:, self.c_index, :]
        loss_dict = {}
        if self.composition:
            nb = self.each_dim[1]
            self.vq_train(gt_poses[:, :nb].contiguous(), 'b', self.g_body, self.g_body_optimizer, loss_dict)
            self.vq_train(gt_poses[:, nb:].contiguous(), 'h', self.g_hand, self.g_hand_optimizer, loss_dict)
        else:
            self.vq_train(gt_poses.contiguous(), 'g', self.g, self.g_optimizer, loss_dict)
        return None, loss_dict

    def vq_train(self, gt, name, model, optimizer, loss_dict):
        bs, n, t = gt.shape
        if next(model.parameters()).device.type != 'cuda':
            model.cuda()
        model.train()
        gt_rows = train_ops.ncl_to_rows(gt)
        optimizer.zero_grad()
        # smplx_body_vq.py:139: encoder, quantizer (EMA codebook update inside, no re-seeding; the decoder sees the OLD
        # codes, vqvae_modules.py:323-334), decoder
        e_q_loss, gt_recon = model(gt)
        if gt_recon.shape[2] != t:
            raise RuntimeError(f"legacy VQ-VAE training needs T divisible by 4 (got {t}: decoder returns {gt_recon.shape[2]})")
        pred = train_ops.to_rows(gt_recon)
        ones = torch.ones(bs * t, device=gt.device)
        rec, sums = train_ops.MaskedL1Losses.apply(pred, gt_rows, ones, float(bs * t), bs, t, [0, n], [1.0], n)
        loss = rec + e_q_loss
        loss.backward()
        optimizer.step()
        model.invalidate()
        s = sums.tolist()
        loss_dict[name + 'rec_loss'] = s[0]
        loss_dict[name + 'velocity_loss'] = s[1]
        loss_dict[name + 'e_q_loss'] = e_q_loss.item()

    def infer_on_audio(self, aud_fn=None, initial_pose=None, norm_stats=None, gt_poses=None, **kwargs):
        """Reconstruct (B,270,T) body+hand poses through the VQ bottleneck (smplx_body_vq.py infer path)."""
        assert self.args.infer, "train mode"
        if self.composition:
            self.g_body.eval(), self.g_hand.eval()
            nb = self.each_dim[1]
            _, body = self.g_body(gt_poses[:, :nb])
            _, hand = self.g_hand(gt_poses[:, nb:])
            return torch.cat([body, hand], dim=1)
        self.g.eval()
        _, rec = self.g(gt_poses)
        return rec
