"""Model registry with the reference's names (nets/__init__.py:2,5-7)."""
from .smplx_body_vq import TrainWrapper as s2g_body_vq
from .predictor import TrainWrapper as s2g_body_predictor
from .refiner import TrainWrapper as s2g_body_refiner
from .vq_teacher import TrainWrapper as s2g_body_vqt
from .base import TrainWrapperBaseClass
from .init_model import init_model
from .ai1 import All_In_One_Model
