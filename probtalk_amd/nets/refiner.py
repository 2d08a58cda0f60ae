"""`s2g_body_refiner` wrapper mirror (nets/inpainting/refiner.py:33-620), inference side."""
import time

import torch
import torch.nn as nn
import torch.optim as optim

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
        super().__init__(args, config)

    def to_parallel(self):
        pass

    def init_optimizer(self):
        self.generator_optimizer = optim.AdamW(
            [{'params': self.RTransformer.parameters()}, {'params': self.AudEnc.parameters()}],
            lr=self.config.Train.learning_rate.generator_learning_rate, betas=[0.9, 0.99])

    def state_dict(self):
        return {'HFTransformer': self.RTransformer.state_dict(), 'AudEnc': self.AudEnc.state_dict(),
                'generator_optim': self.generator_optimizer.state_dict()}

    def load_state_dict(self, state_dict):
        def strip(d):
            return {k.replace('module.', ''): v for k, v in d.items()}
        if 'HFTransformer' in state_dict:
            self.RTransformer.load_state_dict(strip(state_dict['HFTransformer']))
        if 'AudEnc' in state_dict and state_dict['AudEnc']:
            self.AudEnc.load_state_dict(strip(state_dict['AudEnc']))
        if 'generator_optim' in state_dict:
            self.generator_optimizer.load_state_dict(state_dict['generator_optim'])

    def __call__(self, bat):
        raise NotImplementedError("s2g_body_refiner training step is not built yet")

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
