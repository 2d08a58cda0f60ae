"""GPU parity: PQ-VAE / legacy VQ / audio encoder HIP path vs the CPU oracle and the reference's
golden fixtures, all through the C ABI (libprobtalk_hip.so).

Tolerances (BASELINE.json north_star): PQ code indices bit-exact except counted near-ties;
decoded motion within 1e-4 relative error in fp32.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from oracle import probtalk_oracle as O
from probtalk_amd.weights import make_input
from tests.helpers import golden, ref_state_dict, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _vq(device="cuda"):
    from probtalk_amd.nets.modules import VQVAE_SC
    m = VQVAE_SC(376, 512, 128, 1024, 2, 512, q_type="pro", groups=4).eval()
    m.load_state_dict(ref_state_dict("VQ"))
    return m


def _pack(w, kind):
    """reference-layout conv weight -> [k][Npad][Kpad] as runtime.hip packs it."""
    if kind == 4:
        w = w.permute(1, 0, 2)
    N, Cin, kt = w.shape
    Kp, Np = (Cin + 31) // 32 * 32, (N + 127) // 128 * 128
    out = torch.zeros(kt, Np, Kp)
    out[:, :N, :Cin] = w.permute(2, 0, 1)
    return out.contiguous(), Np, Kp


@pytest.mark.parametrize("kind,B,L,Cin,N", [(0, 3, 37, 376, 256), (1, 2, 88, 256, 256), (1, 5, 11, 1024, 1024),
                                            (2, 2, 50, 128, 256), (3, 3, 75, 256, 512), (3, 2, 11, 512, 1024),
                                            (4, 3, 37, 1024, 512), (4, 2, 1, 256, 256), (1, 1, 1, 64, 40)])
def test_conv_op_vs_torch(kind, B, L, Cin, N):
    from probtalk_amd import _lib
    lib = _lib.lib()
    kt = [1, 3, 7, 4, 4][kind]
    g = torch.Generator().manual_seed(kind * 100 + L)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn((Cin, N, kt) if kind == 4 else (N, Cin, kt), generator=g) / (Cin * kt) ** 0.5
    b = torch.randn(N, generator=g)
    if kind == 4:
        ref = F.conv_transpose1d(x, w, b, stride=2, padding=1)
    elif kind == 3:
        ref = F.conv1d(x, w, b, stride=2, padding=1)
    else:
        ref = F.conv1d(x, w, b, padding=kt // 2)
    res = torch.randn(ref.shape, generator=g)
    ref = F.leaky_relu(ref + res, 0.2)
    Lout = ref.shape[2]
    wp, Np, Kp = _pack(w, kind)
    xd = x.cuda()
    a_cl = torch.empty(B * L, Kp, device="cuda")
    _lib.check(lib.pt_ncl_to_cl(_lib.ptr(xd), _lib.ptr(a_cl), B, Cin, L, Kp, _lib.stream()))
    ldc = (N + 31) // 32 * 32
    r_cl = torch.zeros(B * Lout, ldc, device="cuda")
    r_cl[:, :N] = res.permute(0, 2, 1).reshape(B * Lout, N).cuda()
    c_cl = torch.full((B * Lout, ldc), float("nan"), device="cuda")
    wd, bd = wp.cuda(), b.cuda()
    _lib.check(lib.pt_conv1d_cl(_lib.ptr(a_cl), Kp, _lib.ptr(wd), _lib.ptr(bd), N, Np, Kp, kind, B, L, _lib.ptr(c_cl),
                                ldc, 1, 0.2, _lib.ptr(r_cl), ldc, _lib.stream()))
    out = c_cl[:, :N].reshape(B, Lout, N).permute(0, 2, 1).cpu()
    assert rel_err(out, ref) < 1e-5
    assert torch.all(c_cl[:, N:] == 0), "pad columns must be written as zeros"


def test_layout_roundtrip():
    from probtalk_amd import _lib
    lib = _lib.lib()
    x = torch.randn(3, 77, 45, device="cuda")
    cl = torch.empty(3 * 45, 96, device="cuda")
    back = torch.empty_like(x)
    _lib.check(lib.pt_ncl_to_cl(_lib.ptr(x), _lib.ptr(cl), 3, 77, 45, 96, _lib.stream()))
    _lib.check(lib.pt_cl_to_ncl(_lib.ptr(cl), 96, _lib.ptr(back), 3, 77, 45, _lib.stream()))
    assert torch.equal(back, x)
    assert torch.equal(cl[:, :77].reshape(3, 45, 77).permute(0, 2, 1), x)
    assert torch.all(cl[:, 77:] == 0)


def _check_indices(idx_gpu, idx_ref, margins, scale):
    """bit-exact except near-ties: every mismatch must sit on an fp64 margin below 1e-5 * scale."""
    mism = idx_gpu != idx_ref
    n = int(mism.sum())
    if n:
        assert float(margins[mism].max()) < 1e-5 * scale, f"{n} index mismatches are not near-ties"
    return n


def test_pq_search_bit_exact_on_same_latent():
    """quantizer alone on identical z: reference golden (vqvae_modules.py:683-689 shape) and oracle."""
    vq = _vq()
    sd = ref_state_dict("VQ")
    g = golden("pq_b8_w22.npz")
    z = make_input("pq.z", (8, 512, 22)) * 0.5
    qo = vq.vq_layer(z.cuda())
    assert qo.loss.dtype == torch.int64 and tuple(qo.loss.shape) == (8, 22, 4)
    marg = O.pq_margins(sd, "vq_layer.", z, 4)
    n = _check_indices(qo.loss.cpu(), g["indices"], marg, float((z ** 2).sum(1).mean()))
    print(f"pq golden: {n} near-tie mismatches of {g['indices'].numel()}")
    assert n == 0
    assert torch.equal(qo.quantized.cpu(), g["quantized"])
    # large random sweep against the oracle, near-ties counted
    z2 = make_input("pq.z2", (64, 512, 37))
    qo2 = vq.vq_layer(z2.cuda())
    q_ref, idx_ref = O.pq_quantize(sd, "vq_layer.", z2, 4)
    n2 = _check_indices(qo2.loss.cpu(), idx_ref, O.pq_margins(sd, "vq_layer.", z2, 4), float((z2 ** 2).sum(1).mean()))
    print(f"pq sweep: {n2} near-tie mismatches of {idx_ref.numel()}")
    assert n2 <= 2
    same = (qo2.loss.cpu() == idx_ref).all(-1)
    assert torch.equal(qo2.quantized.cpu().permute(0, 2, 1)[same], q_ref.permute(0, 2, 1)[same])


def test_pq_ties_pick_lowest_index():
    from probtalk_amd.nets.modules import ProductQuantization
    pq = ProductQuantization(256, 128, 0.25, 0.99, num_chunks=2).eval()
    cb = torch.randn(128, 128)
    cb[77] = cb[5]          # exact duplicate rows: argmin must return 5
    cb[100] = cb[5]
    pq.embeddings_0.copy_(cb)
    pq.embeddings_1.copy_(cb)
    z = torch.zeros(1, 256, 3)
    z[0, :128, :] = cb[5][:, None]
    z[0, 128:, :] = cb[100][:, None]
    qo = pq(z.cuda())
    assert torch.all(qo.loss.cpu() == 5)


# (256, 88) is BASELINE configs[1] at its full size: 22,528 frames, 2,816 latent rows x 4 sub-codebooks
@pytest.mark.parametrize("B,T,name", [(2, 88, "vq_b2_t88.npz"), (8, 88, None), (3, 300, None), (1, 8, None), (256, 88, None)])
def test_vqvae_encode_decode(B, T, name):
    vq = _vq()
    sd = ref_state_dict("VQ")
    x = make_input("vq.x", (B, 376, T))
    z, qo, _ = vq.encode(x.cuda())
    z_ref, q_ref, idx_ref = O.vqvae_encode(sd, x)
    assert rel_err(z.cpu(), z_ref) < TOL
    marg = O.pq_margins(sd, "vq_layer.", z_ref, 4)
    n = _check_indices(qo.loss.cpu(), idx_ref, marg, float((z_ref ** 2).sum(1).mean()))
    print(f"B={B} T={T}: {n} near-tie index mismatches of {idx_ref.numel()}")
    if name:
        g = golden(name)
        assert rel_err(z.cpu(), g["z"]) < TOL
        assert torch.equal(qo.loss.cpu(), g["indices"])
        assert torch.equal(qo.quantized.cpu(), g["quantized"])
    # decode the ORACLE's latent so the comparison is independent of near-ties
    rec = vq.decode(q_ref.cuda(), None)
    rec_ref = O.vqvae_decode(sd, q_ref)
    assert tuple(rec.shape) == tuple(rec_ref.shape)
    assert rel_err(rec.cpu(), rec_ref) < TOL
    if name:
        assert rel_err(rec.cpu(), g["recon"]) < TOL
    # quantize_all + decode_codes == decode(quantized)
    qa = vq.vq_layer.quantize_all(idx_ref.cuda())
    assert torch.equal(qa.cpu(), O.pq_quantize_all(sd, "vq_layer.", idx_ref))
    rec2 = vq.decode_codes(idx_ref.cuda())
    assert torch.equal(rec2, rec)
    assert torch.equal(vq.decode_codes(idx_ref.to(torch.int32).cuda()), rec)


def test_vqvae_forward_result_forms():
    vq = _vq()
    x = make_input("vq.x", (2, 376, 88)).cuda()
    z, qo, pred = vq(x, result_form="full")
    idx, pred2 = vq(x, result_form="part")
    pred3 = vq(x)
    assert torch.equal(pred, pred2) and torch.equal(pred, pred3) and torch.equal(idx, qo.loss)
    assert tuple(pred.shape) == (2, 376, 88)


def test_vqvae_errors():
    vq = _vq()
    with pytest.raises(RuntimeError):
        vq.encode(torch.zeros(1, 100, 88, device="cuda"))
    with pytest.raises(RuntimeError):
        vq.encode(torch.zeros(1, 376, 4, device="cuda"))       # shorter than the x8 downsampling
    with pytest.raises(NotImplementedError):
        vq.encode(torch.zeros(1, 376, 88, device="cuda"), r=0.1)          # token replacement: off in every config
    # train mode is a real path (tests/test_gpu_module_train.py), not a raise
    vq.cuda().train()
    out = vq(torch.randn(12, 376, 88, device="cuda"))           # 132 latent rows: the EMA re-seeding needs >= 128 (codes)
    assert out.requires_grad and tuple(out.shape) == (12, 376, 88)
    with pytest.raises(RuntimeError):                             # parameters left on the CPU: refused, not dereferenced
        _vq().train()(torch.randn(12, 376, 88, device="cuda"))


def test_legacy_vq():
    from probtalk_amd.nets.modules import VQVAE
    m = VQVAE(270, 64, 128, 1024, 2, 512).eval()
    sd = ref_state_dict("g")
    m.load_state_dict(sd)
    g = golden("legacy_vq_b2_t88.npz")
    x = make_input("lvq.x", (2, 270, 88))
    e, rec = m(x.cuda())
    _, idx = m.encode(x.cuda())
    assert torch.equal(idx.cpu(), g["indices"])
    assert torch.equal(e.cpu(), g["e"])
    assert rel_err(rec.cpu(), g["recon"]) < TOL


def test_condition_encoder():
    from probtalk_amd.nets.modules import ConditionEncoder
    m = ConditionEncoder(768, 256, 2, 256).eval()
    sd = ref_state_dict("AudEnc")
    m.load_state_dict(sd)
    g = golden("predictor_b2_t88.npz")
    aud = make_input("pred.aud", (2, 768, 88))
    out = m(aud.cuda())
    assert rel_err(out.cpu(), g["audio"]) < TOL
    aud2 = make_input("pred.aud300", (3, 768, 300))
    assert rel_err(m(aud2.cuda()).cpu(), O.condition_encoder(sd, aud2)) < TOL


def test_pq_search_resident_kernel_equals_staging_kernel():
    """Large batches take the kernel that keeps a whole sub-codebook in LDS and walks the row tiles (layout_pq.hip);
    batches of fewer than 2 x #CUs tile units take the staging kernel.  Same arithmetic to the bit: the indices and the
    gathered code rows of one 18,944-row call (512 clips x 37 latent steps, ragged last tile included) equal those of
    the same rows searched in 4,096-row pieces."""
    from probtalk_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(77)
    rows, G, K, D = 18944 + 37, 4, 128, 128
    z = torch.randn(rows, G * D, generator=g).cuda()
    cb = (torch.randn(G, K, D, generator=g) * 0.7).cuda().contiguous()
    z[5] = z[4]                                         # a duplicated row: ties inside the batch change nothing
    cb[1, 77] = cb[1, 3]                                # a duplicated code: lowest index must win in both kernels
    idx = torch.full((rows, G), -1, dtype=torch.int64, device="cuda")
    q = torch.full((rows, G * D), float("nan"), device="cuda")
    _lib.check(lib.pt_pq_search_cl(_lib.ptr(z), G * D, _lib.ptr(cb), rows, G, D, K, _lib.ptr(idx), _lib.ptr(q), G * D, _lib.stream()))
    idx2 = torch.full((rows, G), -1, dtype=torch.int64, device="cuda")
    q2 = torch.full((rows, G * D), float("nan"), device="cuda")
    for r0 in range(0, rows, 4096):
        n = min(4096, rows - r0)
        zs, is_, qs = z[r0:r0 + n], idx2[r0:r0 + n], q2[r0:r0 + n]
        _lib.check(lib.pt_pq_search_cl(_lib.ptr(zs), G * D, _lib.ptr(cb), n, G, D, K, _lib.ptr(is_), _lib.ptr(qs), G * D, _lib.stream()))
    torch.cuda.synchronize()
    assert int(idx.min()) >= 0 and int(idx.max()) < K
    assert torch.equal(idx, idx2) and torch.equal(q, q2)
    assert not bool((idx[:, 1] == 77).any()), "a duplicated code must lose to its lower-index twin"
    # and against plain torch on the same operands (near-ties aside)
    d = ((z.view(rows, G, D).double().unsqueeze(2) - cb.double().unsqueeze(0)) ** 2).sum(-1)       # (rows, G, K)
    ref = d.argmin(-1)
    mism = ref != idx
    if int(mism.sum()):
        top2 = d.topk(2, dim=-1, largest=False).values
        assert float((top2[..., 1] - top2[..., 0])[mism].max()) < 1e-5 * float((z ** 2).sum(-1).mean())
