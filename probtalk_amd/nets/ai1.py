"""All_In_One_Model mirror (nets/ai1.py:19-89): chains predictor -> refiner."""
import torch

from .init_model import init_model as _init
from .transformer_modules import interp_blend


def init_model(model_name, model_path, args, config):
    return _init(model_name, args, config, pretrained=True, model_path=model_path)


class All_In_One_Model:
    def __init__(self, face_name, face_path, body_name, body_path, device, args, config):
        self.args = args
        self.config = config
        self.convert_to_6d = config.Data.pose.convert_to_6d
        if face_name is not None:
            raise NotImplementedError("the face model (s2g_face) is outside the token hot path")
        self.face_model = None
        if isinstance(body_name, list):
            self.body_model = [init_model(n, p, args, config) for n, p in zip(body_name, body_path)]
        else:
            self.body_model = init_model(body_name, body_path, args, config)

    def __call__(self, forward_type='infer_on_audio', target_frame=0, result_format='one', **kwargs):
        cost_time = 0
        body_list = []
        if isinstance(self.body_model, list):
            body = None
            for model in self.body_model:
                body, t = getattr(model, forward_type)(**kwargs, target_frame=target_frame, pred_poses=body)
                body_list.append(body)
                cost_time += t
        else:
            body, t = getattr(self.body_model, forward_type)(**kwargs)
            cost_time += t
        holistic_body = body
        if body.shape[2] < target_frame:
            holistic_body = interp_blend(holistic_body, target_frame)
        if result_format == 'one':
            return holistic_body, cost_time
        if body_list[0].shape[2] < target_frame:
            body_list[0] = interp_blend(body_list[0], target_frame)
        return body_list, cost_time
