"""`s2g_body_predictor` wrapper mirror (nets/inpainting/predictor.py:27-686): training step and inference."""
import time

import numpy as np
import torch

from .. import train_ops, training
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
        # data_utils/lower_body.py:43-63: the 135 upper-body / hand joints kept out of SMPL-X's 165, as 6-D pairs
        fix = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 21, 22, 23, 24, 25, 26, 30, 31, 32, 33, 34, 35]
        keep = [i for i in range(165) if i not in fix]
        self.c_index = np.asarray([v for i in keep for v in (2 * i, 2 * i + 1)] if self.convert_to_6d else keep)
        L = self.config.Data.pose.generate_length
        self.random_mask = torch.zeros([L - 29, 1, L])                 # predictor.py:94-96
        for i in range(self.random_mask.shape[0]):
            self.random_mask[i, 0, i:(i + 30)] = 1
        self.train_dropout = 0.1                                        # nn.TransformerDecoderLayer / ConditionEncoder dp
        # dropout seed stream of the training step: each rank draws its own masks (train_ops._Seeds folds the rank in), the
        # position in the stream is checkpointed ('dropout_seeds')
        self._seeds = train_ops._Seeds(0)
        super().__init__(args, config)

    def to_parallel(self):
        pass  # one process per GPU: batch sharding lives in probtalk_amd.dist, not nn.DataParallel

    def init_optimizer(self):
        # torch.optim.AdamW semantics (predictor.py:112-118) on the library's HIP kernel
        self.generator_optimizer = train_ops.AdamW(
            [{'params': self.AudEnc.parameters()}, {'params': self.Predictor.parameters()},
             {'params': self.TextEnc.parameters()}],
            lr=self.config.Train.learning_rate.generator_learning_rate, betas=[0.9, 0.99])

    def state_dict(self):
        return {'AudEnc': self.AudEnc.state_dict(), 'Predictor': self.Predictor.state_dict(),
                'TextEnc': self.TextEnc.state_dict(), 'generator_optim': self.generator_optimizer.state_dict(),
                'dropout_seeds': self._seeds.state_dict()}      # extra key: the reference's loader reads by name

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
        if 'dropout_seeds' in state_dict:
            self._seeds.load_state_dict(state_dict['dropout_seeds'])

    def __call__(self, bat):
        """One training step (predictor.py:168-280): frozen PQ-VAE codes as targets, AudEnc + PredictorNet forward in
        train mode, cross-entropy over all tokens, backward, per-module gradient clipping, AdamW — all on HIP kernels
        (probtalk_amd.training / train_ops).  Optional test hooks in `bat`: '_mask' (B,1,T) and '_keep_scores' (B,N)
        replace the two random draws of the step."""
        self.global_step += 1
        dev = 'cuda'
        aud = bat['aud_feat'].to(dev).to(torch.float32)
        poses = bat['poses'].to(dev).to(torch.float32)
        expression = bat['expression'].to(dev).to(torch.float32)
        id = bat['speaker'].to(dev)
        jaw = poses[:, :self.each_dim[0], :]
        poses = poses[:, self.c_index, :]
        if self.type == 'fbhe':
            poses = torch.cat([jaw, poses, expression], dim=1)
        elif self.type == 'fe':
            poses = torch.cat([jaw, expression], dim=1)
        gt_poses = poses.contiguous()
        bs, n, t = gt_poses.size()
        epoch = bat['epoch']
        if '_mask' in bat:
            mask = bat['_mask'].to(dev).to(torch.float32)
        elif self.motion_context:
            select = torch.randint(0, self.random_mask.shape[0], (bs,))
            mask = self.random_mask[select].clone()
            mask[:64] = 0
            mask = mask.to(dev)
        else:
            mask = torch.zeros([bs, 1, t], device=dev)
        input_poses = gt_poses * mask
        self.VQ.eval()
        with torch.no_grad():
            _, qo_gt, _ = self.VQ.encode(gt_poses=gt_poses)
            state = qo_gt.loss
        epoch_ratio = min(max(epoch / 99, 0.5), 1)
        for m in (self.AudEnc, self.Predictor):
            if next(m.parameters()).device.type != 'cuda':
                m.cuda()
            m.train()
        mods = (self.AudEnc, self.TextEnc, self.Predictor) if self.text else (self.AudEnc, self.Predictor)
        if self.text:
            if next(self.TextEnc.parameters()).device.type != 'cuda':
                self.TextEnc.cuda()
            self.TextEnc.train()
        self._sync_dropout(*mods)
        # one dropout seed stream for the step, consumed in call order (AudEnc, TextEnc, then the decoder layers)
        for m in mods:
            m.__dict__['_pt_seeds'] = self._seeds
        if '_keep_scores' in bat:
            self.Predictor.decoder.__dict__['_pt_keep_scores'] = bat['_keep_scores'].to(dev)
        self.generator_optimizer.zero_grad()
        audio = self.AudEnc(aud)                                                            # predictor.py:243
        text = self.TextEnc(bat['text_feat'].to(dev).to(torch.float32)) if self.text else None          # :248-250
        pred_state = self.Predictor(input_poses, state, mask, id.reshape(-1), audio, text, epoch_ratio)   # :254
        logits = pred_state['logits']
        loss = train_ops.CrossEntropy.apply(logits.reshape(-1, logits.shape[-1]).contiguous(),
                                            state.reshape(-1).to(torch.int64).contiguous())   # :261
        pa, pp = list(self.AudEnc.parameters()), list(self.Predictor.parameters())
        pt = list(self.TextEnc.parameters()) if self.text else []
        multi = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
        if multi and self.__dict__.get('_reducer') is None:
            # one process per GPU in place of nn.DataParallel (to_parallel): mean gradient over ranks, rank 0's BN buffers.
            # Built before backward so that its hooks send each bucket while the earlier layers are still in backward
            from ..dist import GradientReducer
            self._reducer = GradientReducer(pa + pp + pt, list(self.AudEnc.buffers()) + list(self.Predictor.buffers()) +
                                            (list(self.TextEnc.buffers()) if self.text else []))
        loss.backward()
        max_norm = self.config.Train.max_gradient_norm
        if multi:
            self._reducer.reduce()
            self._reducer.sync_buffers()
        grad_ae, coef_ae = train_ops.clip_grad_norm(pa, max_norm)
        grad_p, coef_p = train_ops.clip_grad_norm(pp, max_norm)
        scal = [grad_ae, coef_ae, grad_p, coef_p]
        if pt:
            grad_te, coef_te = train_ops.clip_grad_norm(pt, max_norm)                  # predictor.py:272
            scal += [grad_te, coef_te]
        if multi:
            self._reducer.agree(*scal)
        coefs = {p: coef_ae for p in pa}
        coefs.update({p: coef_p for p in pp})
        if pt:
            coefs.update({p: coef_te for p in pt})
        self.generator_optimizer.step(coefs)
        self.AudEnc.invalidate()          # the packed inference weights are stale now
        self.Predictor.invalidate()
        self.TextEnc.invalidate()
        loss_dict = {'logits': loss, 'loss': loss.item(), 'grad_ae': grad_ae.item(), 'grad_p': grad_p.item()}
        return None, loss_dict

    def _eval(self):
        self.AudEnc.eval(), self.TextEnc.eval(), self.Predictor.eval(), self.VQ.eval()

    def _select(self, gt_poses):
        if self.type == 'bh':
            return gt_poses[:, self.each_dim[0]:-self.each_dim[3]]
        if self.type == 'fe':
            return torch.cat([gt_poses[:, :self.each_dim[0]], gt_poses[:, -self.each_dim[3]:]], 1)
        return gt_poses

    def _infer(self, aud, id, B, gt_poses, mask, noise=None, sync=True, seed=None, text=None):
        gt_poses = self._select(gt_poses)
        aud = aud.repeat(B, 1, 1) if aud.shape[0] != B else aud
        if self.text:
            # predictor.py:415-416: text features repeated per sample and resized to the audio's frame count
            if text is None:
                raise RuntimeError("this predictor was built with p_text: pass the text features")
            text = text.repeat(B, 1, 1) if text.shape[0] != B else text
            text = interp_blend(text, aud.shape[2])
        else:
            text = None
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
        tfeat = self.TextEnc(text) if text is not None else None
        codes = self.Predictor(input_poses, None, mk, id, audio, tfeat, 1, noise=noise, seed=seed)
        pred_poses = self.VQ.decode_codes(codes)                    # quantize_all + transpose + decode
        pred_poses = interp_blend(pred_poses, T, input_poses, mk, premask=False)   # input_poses*mask + pred*(1-mask)
        if sync:
            torch.cuda.synchronize()   # the reference omits this one, so its timings undercount (predictor.py:443)
        return pred_poses, codes, time.time() - start

    def infer_on_batch(self, aud, text, id, B, gt_poses, mask, noise=None, sync=True, seed=None, **kwargs):
        """predictor.py:398-448 -> (poses (B,C,T), seconds).  sync=False enqueues only (seconds ~ 0)."""
        assert self.args.infer, "train mode"
        self._eval()
        poses, _, sec = self._infer(aud, id, B, gt_poses, mask, noise, sync, seed, text=text)
        return poses, sec

    def infer_to_code(self, aud, text, id, B, gt_poses, mask, noise=None, seed=None, **kwargs):
        """predictor.py:597-647 -> (poses, codes (B,T',G), seconds)."""
        assert self.args.infer, "train mode"
        self._eval()
        return self._infer(aud, id, B, gt_poses, mask, noise, seed=seed, text=text)
