"""`s2g_body_vq` wrapper mirror (nets/smplx_body_vq.py:23-307): the legacy single-codebook VQ-VAE
(type 'bh' hard-coded at :37; non-composition form)."""
import torch
import torch.optim as optim

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
        super().__init__(args, config)

    def init_optimizer(self):
        lr = self.config.Train.learning_rate.generator_learning_rate
        if self.composition:
            self.g_body_optimizer = optim.AdamW(self.g_body.parameters(), lr=lr, betas=[0.9, 0.999])
            self.g_hand_optimizer = optim.AdamW(self.g_hand.parameters(), lr=lr, betas=[0.9, 0.999])
        else:
            self.g_optimizer = optim.AdamW(self.g.parameters(), lr=lr, betas=[0.9, 0.999])

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
        raise NotImplementedError("s2g_body_vq training step is not built yet")

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
