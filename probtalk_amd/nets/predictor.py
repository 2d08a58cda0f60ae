"""`s2g_body_predictor` wrapper mirror (nets/inpainting/predictor.py:27-686), inference side."""
import time

import torch
import torch.optim as optim

from .base import TrainWrapperBaseClass, get_speaker_id
from .modules import VQVAE_SC, ConditionEncoder
from .transformer_modules import PredictorNet, interp_blend


class TrainWrapper(TrainWrapperBaseClass):
    def __init__(self, args, config):
        self.args = args
        self.config = config
        self.device = 'cuda'
        self.global_step = 0
        self.convert_to_6d = self.config.Data.pose.convert_to_6d
        self.expression = self.config.Data.pose.expression
        self.epoch = 0
        self.type = self.config.Model.vq_type
        self.init_params()
        self.encoder_choice = self.config.Model.encoder_choice
        self.max_epoch = self.config.Train.epochs
        self.audio = self.config.Model.p_audio
        self.text = self.config.Model.p_text
        self.num_classes = get_speaker_id(config.Data.data_root).__len__()
        self.motion_context = config.Model.motion_context
        if self.type == 'fbhe':
            in_dim = self.full_dim
        elif self.type == 'bh':
            in_dim = self.each_dim[1] + self.each_dim[2]
        elif self.type == 'fe':
            in_dim = self.each_dim[0] + self.each_dim[3]
        emb_dim = config.Model.groups * 128
        self.knn = self.config.Model.knn
        state_dim = emb_dim if self.knn else config.Model.code_num
        aud_dim, text_dim = self.getFM_dim(self.audio, self.text)
        self.AudEnc = ConditionEncoder(in_dim=aud_dim, num_hiddens=256, num_residual_layers=2, num_residual_hiddens=256)
        self.TextEnc = ConditionEncoder(in_dim=text_dim, num_hiddens=256, num_residual_layers=2,
                                        num_residual_hiddens=256)
        self.Predictor = PredictorNet(self.knn, in_dim, state_dim, 512, 10, 6, self.num_classes,
                                      groups=config.Model.groups, identity=config.Model.identity,
                                      maskgit=config.Model.maskgit, maskgit_T=config.Model.maskgit_T,
                                      transformer=config.Model.transformer, text=self.text, audio=self.audio,
                                      motion_context=config.Model.motion_context)
        self.VQ = VQVAE_SC(in_dim, embedding_dim=emb_dim, num_embeddings=config.Model.code_num, num_hiddens=1024,
                           num_residual_layers=2, num_residual_hiddens=512, groups=config.Model.groups,
                           q_type=config.Model.q_type, share_code=config.Model.share_code)
        model_ckpt = torch.load(self.config.Model.vq_path, map_location=torch.device('cpu'), weights_only=True)
        self.VQ.load_state_dict(model_ckpt['generator']['VQ'])
        self.in_dim = in_dim
        self.discriminator = None
        super().__init__(args, config)

    def to_parallel(self):
        pass  # one process per GPU: batch sharding lives in probtalk_amd.dist, not nn.DataParallel

    def init_optimizer(self):
        self.generator_optimizer = optim.AdamW(
            [{'params': self.AudEnc.parameters()}, {'params': self.Predictor.parameters()},
             {'params': self.TextEnc.parameters()}],
            lr=self.config.Train.learning_rate.generator_learning_rate, betas=[0.9, 0.99])

    def state_dict(self):
        return {'AudEnc': self.AudEnc.state_dict(), 'Predictor': self.Predictor.state_dict(),
                'TextEnc': self.TextEnc.state_dict(), 'generator_optim': self.generator_optimizer.state_dict()}

    def load_state_dict(self, state_dict):
        def strip(d):
            return {k.replace('module.', ''): v for k, v in d.items()}
        if 'AudEnc' in state_dict:
            self.AudEnc.load_state_dict(strip(state_dict['AudEnc']))
        if 'Predictor' in state_dict:
            self.Predictor.load_state_dict(strip(state_dict['Predictor']))
        if 'TextEnc' in state_dict:
            self.TextEnc.load_state_dict(strip(state_dict['TextEnc']))
        if 'generator_optim' in state_dict:
            self.generator_optimizer.load_state_dict(state_dict['generator_optim'])

    def __call__(self, bat):
        raise NotImplementedError("s2g_body_predictor training step (fwd+bwd+AdamW) is not built yet")

    def _eval(self):
        self.AudEnc.eval(), self.TextEnc.eval(), self.Predictor.eval(), self.VQ.eval()

    def _select(self, gt_poses):
        if self.type == 'bh':
            return gt_poses[:, self.each_dim[0]:-self.each_dim[3]]
        if self.type == 'fe':
            return torch.cat([gt_poses[:, :self.each_dim[0]], gt_poses[:, -self.each_dim[3]:]], 1)
        return gt_poses

    def _infer(self, aud, id, B, gt_poses, mask, noise=None, sync=True):
        gt_poses = self._select(gt_poses)
        aud = aud.repeat(B, 1, 1) if aud.shape[0] != B else aud
        id = torch.as_tensor(id).to('cuda').reshape(-1)
        id = id.repeat(B) if id.numel() == 1 else id
        T = aud.shape[2]
        gt = gt_poses.expand(B, -1, -1)
        mk = mask.expand(B, -1, -1)
        if sync:
            torch.cuda.synchronize()
        start = time.time()
        input_poses = interp_blend(None, T, gt, mk, premask=False)  # gt_poses * mask
        audio = self.AudEnc(aud)
        codes = self.Predictor(input_poses, None, mk, id, audio, None, 1, noise=noise)
        pred_poses = self.VQ.decode_codes(codes)                    # quantize_all + transpose + decode
        pred_poses = interp_blend(pred_poses, T, input_poses, mk, premask=False)   # input_poses*mask + pred*(1-mask)
        if sync:
            torch.cuda.synchronize()   # the reference omits this one, so its timings undercount (predictor.py:443)
        return pred_poses, codes, time.time() - start

    def infer_on_batch(self, aud, text, id, B, gt_poses, mask, noise=None, sync=True, **kwargs):
        """predictor.py:398-448 -> (poses (B,C,T), seconds).  sync=False enqueues only (seconds ~ 0)."""
        assert self.args.infer, "train mode"
        self._eval()
        poses, _, sec = self._infer(aud, id, B, gt_poses, mask, noise, sync)
        return poses, sec

    def infer_to_code(self, aud, text, id, B, gt_poses, mask, noise=None, **kwargs):
        """predictor.py:597-647 -> (poses, codes (B,T',G), seconds)."""
        assert self.args.infer, "train mode"
        self._eval()
        return self._infer(aud, id, B, gt_poses, mask, noise)
