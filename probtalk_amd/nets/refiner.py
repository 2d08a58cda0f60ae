"""`s2g_body_refiner` wrapper mirror (nets/inpainting/refiner.py:33-620)."""
import time

import numpy as np
import torch
import torch.nn as nn

from .. import train_ops, training
from .base import TrainWrapperBaseClass
from .modules import VQVAE_SC
from .transformer_modules import RTransformer, interp_blend


class TrainWrapper(TrainWrapperBaseClass):
    def __init__(self, args, config):
        self.args = args
        self.config = config
        self.device = torch.device(self.args.gpu)
        self.global_step = 0
        self.convert_to_6d = self.config.Data.pose.convert_to_6d
        self.expression = self.config.Data.pose.expression
        self.epoch = 0
        self.type = self.config.Model.hf_type
        self.vqtype = self.config.Model.hf_vq_type
        self.init_params()
        self.audio = True
        self.composition = self.config.Model.composition
        self.bh_model = self.config.Model.bh_model
        self.two_stage = self.config.Model.two_stage
        self.motion_context = self.config.Model.motion_context
        dims = {'fbhe': self.full_dim, 'bh': self.each_dim[1] + self.each_dim[2],
                'fe': self.each_dim[0] + self.each_dim[3]}
        in_dim = dims[self.type]
        emb_dim = config.Model.groups * 128
        self.knn = self.config.Model.knn
        self.RTransformer = RTransformer(in_dim, 768, 512, 6, use_label=config.Model.identity,
                                         motion_context=self.motion_context)
        self.AudEnc = nn.Identity()
        vq_dim = dims.get(self.vqtype, 0)
        if self.vqtype != 'novq':
            self.VQ = VQVAE_SC(vq_dim, embedding_dim=emb_dim, q_type=config.Model.q_type,
                               num_embeddings=config.Model.code_num, num_hiddens=1024, num_residual_layers=2,
                               num_residual_hiddens=512, groups=config.Model.groups,
                               share_code=config.Model.share_code)
            model_ckpt = torch.load(self.config.Model.vq_path, map_location=torch.device('cpu'), weights_only=True)
            self.VQ.load_state_dict(model_ckpt['generator']['VQ'])
        else:
            self.VQ = nn.Identity()
        self.in_dim = in_dim
        self.vq_dim = vq_dim
        self.discriminator = None
        fix = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 21, 22, 23, 24, 25, 26, 30, 31, 32, 33, 34, 35]
        keep = [i for i in range(165) if i not in fix]        # data_utils/lower_body.py:43-63
        self.c_index = np.asarray([v for i in keep for v in (2 * i, 2 * i + 1)] if self.convert_to_6d else keep)
        L = self.config.Data.pose.generate_length
        self.random_mask = torch.zeros([L - 29, 1, L])         # refiner.py:103-106
        for i in range(self.random_mask.shape[0]):
            self.random_mask[i, 0, i:(i + 30)] = 1
        self.train_dropout = 0.1
        self._seeds = train_ops._Seeds(1)                         # per-rank dropout masks (rank folded in); position checkpointed
        super().__init__(args, config)

    def to_parallel(self):
        pass

    def init_optimizer(self):
        self.generator_optimizer = train_ops.AdamW(
            [{'params': self.RTransformer.parameters()}, {'params': self.AudEnc.parameters()}],
            lr=self.config.Train.learning_rate.generator_learning_rate, betas=[0.9, 0.99])

    def state_dict(self):
        return {'HFTransformer': self.RTransformer.state_dict(), 'AudEnc': self.AudEnc.state_dict(),
                'generator_optim': self.generator_optimizer.state_dict(), 'dropout_seeds': self._seeds.state_dict()}

    def load_state_dict(self, state_dict):
        def strip(d):
            return {k.replace('module.', ''): v for k, v in d.items()}
        if 'HFTransformer' in state_dict:
            self.RTransformer.load_state_dict(strip(state_dict['HFTransformer']))
        if 'AudEnc' in state_dict and state_dict['AudEnc']:
            self.AudEnc.load_state_dict(strip(state_dict['AudEnc']))
        if 'generator_optim' in state_dict:
            self.generator_optimizer.load_state_dict(state_dict['generator_optim'])
        if 'dropout_seeds' in state_dict:
            self._seeds.load_state_dict(state_dict['dropout_seeds'])

    def __call__(self, bat):
        """One refiner training step (refiner.py:183-311) on the HIP training kernels: frozen PQ-VAE reconstruction of
        the ground truth (+ 0.05 N(0,1) augmentation) as the coarse input, 30-frame context mask, RTransformer train
        forward, masked part-L1 + velocity loss, backward, clipping, AdamW.  Test hooks in `bat`: '_mask' (B,1,T),
        '_noise' (B,C,T) replace the two random draws."""
        self.global_step += 1
        if self.type != 'fbhe' or self.vqtype != 'fbhe':
            raise NotImplementedError("refiner training is built for hf_type = hf_vq_type = 'fbhe' (config/transformer.json)")
        dev = 'cuda'
        aud = bat['aud_feat'].to(dev).to(torch.float32)
        poses = bat['poses'].to(dev).to(torch.float32)
        expression = bat['expression'].to(dev).to(torch.float32)
        id = bat['speaker'].to(dev)
        jaw = poses[:, :self.each_dim[0], :]
        gt_poses = torch.cat([jaw, poses[:, self.c_index, :], expression], dim=1).contiguous()
        bs, n, t = gt_poses.size()
        if '_mask' in bat:
            mask = bat['_mask'].to(dev).to(torch.float32)
        elif self.motion_context:
            select = torch.randint(0, self.random_mask.shape[0], (bs,))
            mask = self.random_mask[select].clone()
            mask[:64] = 0
            mask = mask.to(dev)
        else:
            mask = torch.zeros([bs, 1, t], device=dev)
        with torch.no_grad():
            self.VQ.eval()
            vq_poses = self.VQ(gt_poses=gt_poses)
            noise = bat['_noise'].to(dev) if '_noise' in bat else torch.randn(bs, n, t, device=dev)
            # F.interpolate(vq, t) + 0.05*noise, then gt*mask + (.)*(1-mask)  (refiner.py:247,267-270)
            coarse = interp_blend(vq_poses, t, torch.zeros_like(gt_poses), torch.zeros_like(mask), premask=False)
            input_poses = gt_poses * mask + (coarse + 0.05 * noise) * (1 - mask)
        rt = self.RTransformer
        if next(rt.parameters()).device.type != 'cuda':
            rt.cuda()
        rt.train()
        self._sync_dropout(rt)
        rt.__dict__['_pt_seeds'] = self._seeds
        self.generator_optimizer.zero_grad()
        pred_poses = rt(input_poses, aud, mask, id.reshape(-1))                              # refiner.py:296
        nmask = (1.0 - mask).reshape(-1).contiguous()
        loss, sums = train_ops.MaskedL1Losses.apply(train_ops.to_rows(pred_poses), train_ops.ncl_to_rows(gt_poses), nmask,
                                                    float(nmask.sum()), bs, t, [0, 6, 96, 276, 376], [0.25] * 4, 276)   # :297
        params = list(rt.parameters())
        multi = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
        if multi and self.__dict__.get('_reducer') is None:
            from ..dist import GradientReducer         # before backward: its hooks overlap the all-reduce with it
            self._reducer = GradientReducer(params, list(rt.buffers()))
        loss.backward()
        if multi:
            self._reducer.reduce()
        grad_HF, coef = train_ops.clip_grad_norm(params, self.config.Train.max_gradient_norm)
        if multi:
            self._reducer.agree(grad_HF, coef)
        self.generator_optimizer.step({p: coef for p in params})
        rt.invalidate()
        s = sums.tolist()
        loss_dict = {'face_loss': s[0], 'body_loss': s[1], 'rec_loss': (s[0] + s[1] + s[2] + s[3]) / 4, 'velocity_loss': s[4],
                     'loss': loss.item(), 'grad_HF': grad_HF.item(), 'grad_A': 0.0}
        return None, loss_dict

    def infer_on_batch(self, aud, B, id, gt_poses, mask, pred_poses=None, sync=True, **kwargs):
        """refiner.py:512-558 -> (poses (B,C,T), seconds)."""
        assert self.args.infer, "train mode"
        self.RTransformer.eval(), self.AudEnc.eval(), self.VQ.eval()
        if self.type != 'fbhe':
            raise NotImplementedError("refiner hf_type other than 'fbhe' is not on the configured path")
        aud = aud.repeat(B, 1, 1) if aud.shape[0] != B else aud
        T = aud.shape[2]
        gt = gt_poses.expand(B, -1, -1)
        mk = mask.expand(B, -1, -1)
        id = torch.as_tensor(id).to(aud.device).reshape(-1)
        if self.two_stage:
            input_poses = interp_blend(pred_poses, T, gt, mk, premask=False)   # gt*mask + interp(pred)*(1-mask)
        else:
            input_poses = interp_blend(torch.zeros_like(gt), T, gt, mk, premask=False)
        if sync:
            torch.cuda.synchronize()
        start = time.time()
        aud_feat = self.AudEnc(aud)
        out = self.RTransformer(input_poses, aud_feat, mk, id)
        out = interp_blend(out, T, gt, mk, premask=False)                      # gt*mask + pred*(1-mask)
        if sync:
            torch.cuda.synchronize()
        return out, time.time() - start
