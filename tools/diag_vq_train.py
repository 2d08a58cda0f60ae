import math, torch
from oracle import probtalk_oracle_train as OT
from oracle import probtalk_oracle as O
from probtalk_amd import train_ops as T, training
from tests.helpers import ref_state_dict
from tests.test_oracle_vq_train_golden import vq_train_case
from tests.test_gpu_vq_train import _cuda_params
sd = ref_state_dict("VQ")
x = vq_train_case()
B, _, Tn = x.shape
pre = []
orig = O._leaky
def rec(v, slope):
    if slope > 0: pre.append(v.detach())
    return orig(v, slope)
O._leaky = rec
loss_o, parts_o, grads_o, new_o, pred_o, idx_o = OT.vq_train_grads(sd, x)
O._leaky = orig
P = _cuda_params(sd)
cb = torch.stack([P[f"vq_layer.embeddings_{i}"] for i in range(4)]).contiguous()
x_rows = T.ncl_to_rows(x.cuda())
ys = []
obn = T.bn_train
def bn(x_, g, b, rm, rv, slope=-1.0, res=None, momentum=0.1):
    y = obn(x_, g, b, rm, rv, slope=slope, res=res, momentum=momentum)
    if slope > 0: ys.append(y.detach())
    return y
T.bn_train = bn
z, e_q, idx, pred, Tout = training.vqvae_train_forward(P, cb, x_rows, B, Tn)
T.bn_train = obn
print(len(pre), len(ys))
for i, (p, y) in enumerate(zip(pre, ys)):
    Bc, C, L = p.shape
    yn = y.cpu().reshape(Bc, L, -1)[:, :, :C].permute(0, 2, 1)
    ref = torch.where(p > 0, p, 0.2 * p)
    differ = (yn > 0) != (p > 0)
    print(i, tuple(p.shape), "flips", int(differ.sum()), "max|pre| at flips", float(p[differ].abs().max()) if differ.any() else 0.0,
          "fwd err", float((yn - ref).abs().max() / ref.abs().max()), "n(|pre|<1e-5)", int((p.abs() < 1e-5).sum()))
