"""TrainWrapperBaseClass mirror (nets/base.py:11-130): dimension bookkeeping and shared plumbing."""
import os

import torch.optim as optim

expression_dim = int(os.environ.setdefault('expression_dim', '100'))   # data_utils/consts.py:3

# data_utils/consts.py:6-11
speaker_id = {'oliver': 0, 'chemistry': 1, 'seth': 2, 'conan': 3}


def get_speaker_id(data_root):
    """data_utils/consts.py:13-18: one id per entry of the dataset root.  The SHOW dataset is not
    present on synthetic runs; then the four known speakers are used."""
    if data_root and os.path.isdir(data_root):
        return {k: i for i, k in enumerate(os.listdir(data_root))}
    return dict(speaker_id)


class TrainWrapperBaseClass():
    def __init__(self, args, config) -> None:
        self.init_optimizer()
        self.num_classes = get_speaker_id(config.Data.data_root).__len__()

    def init_optimizer(self) -> None:
        self.generator_optimizer = optim.Adam(self.generator.parameters(),
                                              lr=self.config.Train.learning_rate.generator_learning_rate,
                                              betas=[0.9, 0.999])

    def to_parallel(self):
        pass

    def _sync_dropout(self, *modules):
        """Wrappers with a `train_dropout` attribute (tests switch dropout off to compare with fixtures): every
        nn.Dropout the module was built with p > 0 runs at that rate."""
        import torch.nn as nn
        for m in modules:
            for sub in m.modules():
                if isinstance(sub, nn.Dropout):
                    if '_pt_p0' not in sub.__dict__:
                        sub.__dict__['_pt_p0'] = sub.p
                    sub.p = self.train_dropout if sub.__dict__['_pt_p0'] > 0 else 0.0

    def __call__(self, bat):
        raise NotImplementedError

    def init_params(self):
        scale = 2 if self.convert_to_6d else 1
        jaw_dim = round(3 * scale)
        eye_dim = 0
        body_dim = round(45 * scale)
        hand_dim = 2 * round(45 * scale)
        face_dim = expression_dim
        b_j = 0
        b_e = b_j + jaw_dim
        b_b = b_e + eye_dim
        b_h = b_b + body_dim
        b_f = b_h + hand_dim
        self.dim_list = [b_j, b_e, b_b, b_h, b_f]
        self.full_dim = jaw_dim + eye_dim + body_dim + hand_dim + face_dim
        self.pose = int(self.full_dim / round(3 * scale))
        self.each_dim = [jaw_dim, eye_dim + body_dim, hand_dim, face_dim]

    def getFM_dim(self, audio_name, text_name):
        aud = {'hubert': 1024, 'wav2vec': 768, 'speech2text': 768, None: 768}
        txt = {'gpt2': 768, 'clip': 512, 'bert': 768, 't5': 768, 'ton': 3, None: 768}
        if audio_name not in aud:
            raise NameError("The audio model name is incorrect.")
        if text_name == 'fasttext':
            raise NotImplementedError
        if text_name not in txt:
            raise NameError("The text model name is incorrect.")
        return aud[audio_name], txt[text_name]

    @staticmethod
    def _strip_module_prefix(state_dict):
        out = {}
        for k, v in state_dict.items():
            out[k] = None if v is None else {k1.replace('module.', ''): v1 for k1, v1 in v.items()} \
                if isinstance(v, dict) and k != 'generator_optim' and not k.endswith('_optim') else v
        return out
