"""Write reference-layout checkpoint files (ckpt-99.pth for s2g_body_vqt / s2g_body_predictor / s2g_body_refiner) from
generator weights: what tests/test_gpu_checkpoint.py loads through All_In_One_Model and init_model(pretrained=True).

    python tools/write_reference_ckpts.py <outdir> [seed]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probtalk_amd.synthetic import write_reference_checkpoints  # noqa: E402

if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "ckpts"
    for k, p in write_reference_checkpoints(out, int(sys.argv[2]) if len(sys.argv) > 2 else 0).items():
        print(k, p, os.path.getsize(p), "bytes")
