"""JSON config -> attribute tree, as the reference's trainer/config.py:12-18,43-47 does."""
import json


class Object:
    def __init__(self, config: dict):
        for k, v in config.items():
            setattr(self, k, Object(v) if isinstance(v, dict) else v)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Object) else v) for k, v in self.__dict__.items()}


def load_JsonConfig(json_file):
    with open(json_file, "r") as f:
        return Object(json.load(f))


def default_transformer_config(vq_path="", data_root=""):
    """The values of the reference's config/transformer.json that the hot path reads."""
    return Object({
        "Data": {"data_root": data_root,
                 "pose": {"convert_to_6d": True, "expression": True, "generate_length": 180, "normalization": True}},
        "Model": {"model_name": "s2g_body_refiner", "p_audio": "wav2vec", "p_text": None, "two_stage": True,
                  "hf_type": "fbhe", "hf_vq_type": "fbhe", "motion_context": True, "transformer": True,
                  "identity": True, "maskgit": True, "maskgit_T": 8, "composition": True, "code_num": 128,
                  "code_dim": 128, "share_code": False, "vq_type": "fbhe", "q_type": "pro", "bh_model": True,
                  "groups": 4, "knn": False, "encoder_choice": "wav2vec", "vq_path": vq_path},
        "Train": {"epochs": 100, "max_gradient_norm": 5,
                  "learning_rate": {"generator_learning_rate": 1e-4, "discriminator_learning_rate": 1e-4}},
    })
