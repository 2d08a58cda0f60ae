import torch
from oracle import probtalk_oracle_train as OT
from probtalk_amd import training
from probtalk_amd.weights import make_input
from tests.helpers import ref_state_dict
from tests.test_oracle_train_golden import train_case
from tests.test_gpu_train import _cuda_params
sa, sp = ref_state_dict("AudEnc"), ref_state_dict("Predictor")
for trial in range(4):
    aud, inp, mask, label, state, keep, ratio = train_case()
    if trial:
        aud = make_input(f"diag.aud{trial}", aud.shape)
        inp = make_input(f"diag.gt{trial}", inp.shape) * mask
    if trial == 3:
        sa = {k: v.double() if v.dtype == torch.float32 else v for k, v in sa.items()}
        sp = {k: v.double() if v.dtype == torch.float32 else v for k, v in sp.items()}
        aud, inp, mask, keep = aud.double(), inp.double(), mask.double(), keep.double()
    loss_o, grads_o, running_o, logits_o = OT.predictor_train_grads(sa, sp, aud, inp, mask, label, state, keep, ratio)
    if trial == 3:
        sa, sp = ref_state_dict("AudEnc"), ref_state_dict("Predictor")
        aud, inp, mask, keep = aud.float(), inp.float(), mask.float(), keep.float()
    Pa, Pp = _cuda_params(sa), _cuda_params(sp)
    loss, logits = training.predictor_loss(Pa, Pp, aud.cuda(), inp.cuda(), mask.cuda(), label.cuda(), state.cuda(), keep.cuda(), ratio, p_drop=0.0, seeds=None, groups=4)
    loss.backward()
    rows = []
    for (tag, k), gref in grads_o.items():
        got = (Pa if tag == "AudEnc" else Pp)[k].grad
        if got is None: continue
        d = (got.cpu().double() - gref.double()).abs()
        m = float(gref.abs().max())
        if m < 1e-6: continue
        rows.append((float(d.max()) / m, tag, k, int((d > 1e-3 * m).sum()), d.numel()))
    rows.sort(reverse=True)
    print("trial", trial, float(loss), float(loss_o))
    for r in rows[:5]:
        print("  ", r)
