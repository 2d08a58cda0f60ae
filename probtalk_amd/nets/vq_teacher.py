"""`s2g_body_vqt` wrapper mirror (nets/inpainting/vq_teacher.py:37-285)."""
import time

import numpy as np
import torch

from .. import train_ops, training
from .base import TrainWrapperBaseClass
from .modules import VQVAE_SC


class TrainWrapper(TrainWrapperBaseClass):
    def __init__(self, args, config):
        self.args = args
        self.config = config
        self.device = torch.device(self.args.gpu)
        self.global_step = 0
        self.convert_to_6d = self.config.Data.pose.convert_to_6d
        self.expression = self.config.Data.pose.expression
        self.epoch = 0
        self.type = self.config.Model.vq_type
        self.init_params()
        emb_dim = config.Model.groups * config.Model.code_dim
        if self.type == 'fbhe':
            in_dim = self.full_dim
        elif self.type == 'bh':
            in_dim = self.each_dim[1] + self.each_dim[2]
        elif self.type == 'fe':
            in_dim = self.each_dim[0] + self.each_dim[3]
        self.VQ = VQVAE_SC(in_dim, embedding_dim=emb_dim, num_embeddings=config.Model.code_num, num_hiddens=1024,
                           num_residual_layers=2, num_residual_hiddens=512, q_type=config.Model.q_type,
                           groups=config.Model.groups, share_code=config.Model.share_code)
        self.discriminator = None
        self.in_dim = in_dim
        # data_utils/lower_body.py:43-63 (see predictor.py)
        fix = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 21, 22, 23, 24, 25, 26, 30, 31, 32, 33, 34, 35]
        keep = [i for i in range(165) if i not in fix]
        self.c_index = np.asarray([v for i in keep for v in (2 * i, 2 * i + 1)] if self.convert_to_6d else keep)
        super().__init__(args, config)

    def init_optimizer(self):
        # optim.Adam(betas 0.9/0.999) (vq_teacher.py:97-101) = the AdamW kernel with weight decay 0
        self.generator_optimizer = train_ops.AdamW([{'params': self.VQ.parameters()}],
                                                   lr=self.config.Train.learning_rate.generator_learning_rate,
                                                   betas=[0.9, 0.999], weight_decay=0.0)

    def state_dict(self):
        return {'VQ': self.VQ.state_dict(), 'generator_optim': self.generator_optimizer.state_dict()}

    def load_state_dict(self, state_dict):
        if 'VQ' in state_dict:
            self.VQ.load_state_dict({k.replace('module.', ''): v for k, v in state_dict['VQ'].items()})
        if 'generator_optim' in state_dict:
            self.generator_optimizer.load_state_dict(state_dict['generator_optim'])

    def _loss_spec(self):
        """(column bounds, weights, velocity columns) of get_loss (vq_teacher.py:191-226) per vq_type."""
        d, e = self.dim_list, self.each_dim
        if self.type == 'fbhe':
            return [0, d[1], d[3], d[4], self.in_dim], [0.5 - 0.4, 0.4, 0.4, 0.5 - 0.4], ('face_loss', 'body_loss', 'hand_loss', 'exp_loss')
        if self.type == 'bh':
            return [0, e[1], self.in_dim], [0.5, 0.5], ('body_loss', 'hand_loss')
        return [0, d[1] - d[0], self.in_dim], [0.5, 0.5], ('jaw_loss', 'exp_loss')

    def __call__(self, bat):
        """One PQ-VAE training step (vq_teacher.py:135-182) on the HIP training kernels: encoder / decoder with
        batch-statistic BatchNorm, nearest-code search, commitment loss + straight-through estimator, EMA codebook
        update with dead-code re-seeding, part-weighted smooth-L1 + velocity losses, backward, Adam.
        The reference clips gradients BEFORE loss.backward() (173 vs 177): the call sees zeroed/absent gradients, so it
        never scales anything and `grad` is always 0; that behaviour is kept."""
        self.global_step += 1
        dev = 'cuda'
        poses = bat['poses'].to(dev).to(torch.float32)
        expression = bat['expression'].to(dev).to(torch.float32)
        jaw = poses[:, :self.each_dim[0], :]
        poses = poses[:, self.c_index, :]
        if self.type == 'fbhe':
            poses = torch.cat([jaw, poses, expression], dim=1)
        elif self.type == 'fe':
            poses = torch.cat([jaw, expression], dim=1)
        gt_poses = poses.contiguous()
        bs, n, t = gt_poses.size()
        vq = self.VQ
        if next(vq.parameters()).device.type != 'cuda':
            vq.cuda()
        vq.train()
        gt_rows = train_ops.ncl_to_rows(gt_poses)
        self.generator_optimizer.zero_grad()
        # vq_teacher.py:160: encoder, quantizer (EMA codebook update inside; the decoder sees the OLD codes), decoder
        z, qo, pred_poses = vq(gt_poses=gt_poses, result_form='full')
        e_q_loss = qo.loss
        if pred_poses.shape[2] != t:
            raise RuntimeError(f"PQ-VAE training needs T divisible by 8 (got {t}: decoder returns {pred_poses.shape[2]})")
        pred = train_ops.to_rows(pred_poses)
        bounds, weights, names = self._loss_spec()
        vel_cols = n - self.each_dim[3]
        rec, sums = train_ops.VQLosses.apply(pred, gt_rows, bs, t, bounds, weights, vel_cols, 0.01)
        loss = rec + e_q_loss
        loss.backward()
        self.generator_optimizer.step()
        vq.invalidate()
        s = sums.tolist()
        loss_dict = {k: s[i] for i, k in enumerate(names)}
        loss_dict.update({'velocity_loss': s[len(names)], 'e_q_loss': e_q_loss.item(), 'q_loss': 0, 'lr_loss': 0, 'grad': 0.0})
        return None, loss_dict

    def infer_on_audio(self, gt_poses, aud, text, id, B, **kwargs):
        """vq_teacher.py:255-275: encode -> quantize -> decode, (poses, seconds)."""
        assert self.args.infer, "train mode"
        self.VQ.eval()
        if gt_poses.shape[1] in (238, 376) and self.type == 'bh':
            gt_poses = gt_poses[:, self.each_dim[0]:-self.each_dim[3]]
        torch.cuda.synchronize()
        start = time.time()
        _, qo, _ = self.VQ.encode(gt_poses)
        pred_poses = self.VQ.decode(qo.quantized, None)
        torch.cuda.synchronize()
        return pred_poses, time.time() - start

    # the reference's infer_on_batch unpacks a QuantizationOutput as a tuple (vq_teacher.py:250) and cannot run;
    # the working path is the one above
    infer_on_batch = infer_on_audio

    def get_latents(self, gt_poses=None, **kwargs):
        self.VQ.eval()
        _, qo, _ = self.VQ.encode(gt_poses)
        return qo.loss
