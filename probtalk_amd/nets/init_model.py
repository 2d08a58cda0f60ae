"""init_model mirror (nets/init_model.py:6-12)."""
import torch


def init_model(model_name, args, config, pretrained=False, model_path=None):
    from .. import nets
    generator = getattr(nets, model_name)(args, config)
    if pretrained:
        model_ckpt = torch.load(model_path, map_location=torch.device('cpu'), weights_only=True)
        generator.load_state_dict(model_ckpt['generator'])
    return generator
