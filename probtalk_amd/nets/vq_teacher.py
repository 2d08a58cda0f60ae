"""`s2g_body_vqt` wrapper mirror (nets/inpainting/vq_teacher.py:37-285), inference side."""
import time

import torch
import torch.optim as optim

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
        super().__init__(args, config)

    def init_optimizer(self):
        self.generator_optimizer = optim.Adam([{'params': self.VQ.parameters()}],
                                              lr=self.config.Train.learning_rate.generator_learning_rate,
                                              betas=[0.9, 0.999])

    def state_dict(self):
        return {'VQ': self.VQ.state_dict(), 'generator_optim': self.generator_optimizer.state_dict()}

    def load_state_dict(self, state_dict):
        if 'VQ' in state_dict:
            self.VQ.load_state_dict({k.replace('module.', ''): v for k, v in state_dict['VQ'].items()})
        if 'generator_optim' in state_dict:
            self.generator_optimizer.load_state_dict(state_dict['generator_optim'])

    def __call__(self, bat):
        raise NotImplementedError("s2g_body_vqt training step (EMA codebook update, losses) is not built yet")

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
