"""GPU parity of the phase-pipelined LDS-DMA tap-GEMM for the bf16 conv stacks (conv8_kernel, gemm_bf16_conv.hip) through
pt_conv1d_cl_bf16: every conv form of the path (1x1, k3, k7, k4 s2, ConvTranspose k4 s2) with bias, LeakyReLU / ReLU and
the bf16 residual, against torch's conv in float64 on the same bf16-rounded operands, and against the 4-wave kernel it
replaces (option conv8 = 0): same products, fp32 accumulation in a different order."""
import pytest
import torch
import torch.nn.functional as F

from probtalk_amd import _lib
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _pack(w, kind):
    """reference-layout conv weight -> [k][Npad][Kpad] bf16 as runtime.hip packs it (Kpad multiple of 64)."""
    if kind == 4:
        w = w.permute(1, 0, 2)
    N, Cin, kt = w.shape
    Kp, Np = (Cin + 63) // 64 * 64, (N + 127) // 128 * 128
    out = torch.zeros(kt, Np, Kp)
    out[:, :N, :Cin] = w.permute(2, 0, 1)
    return out.to(torch.bfloat16).contiguous(), Np, Kp


# (kind, B, L, Cin, N, slope, residual): B*Lout/256 * ceil(N/256) >= 64 tiles so that the new kernel takes the launch;
# lengths that are not multiples of anything, clips shorter than a tile (several clips per 256-row tile), N = 376 (ragged
# last column tile), Cin = 384 (6 channel blocks)
CASES = [(1, 60, 300, 256, 256, 0.2, False), (1, 59, 297, 256, 256, 0.0, True), (1, 450, 37, 1024, 1024, 0.2, True),
         (0, 57, 300, 256, 376, -1.0, False), (0, 64, 300, 768, 256, 0.1, False), (3, 120, 300, 256, 256, 0.2, False),
         (3, 130, 149, 384, 512, 0.2, False), (4, 230, 74, 512, 256, 0.2, False), (4, 470, 37, 1024, 512, 0.2, False),
         (2, 70, 300, 128, 256, 0.2, False)]


@pytest.mark.parametrize("kind,B,L,Cin,N,slope,with_res", CASES)
def test_conv8_matches_torch_and_the_4wave_kernel(kind, B, L, Cin, N, slope, with_res):
    lib = _lib.lib()
    kt = [1, 3, 7, 4, 4][kind]
    g = torch.Generator().manual_seed(kind * 1000 + L + N)
    x = torch.randn(B, Cin, L, generator=g).to(torch.bfloat16)
    w = (torch.randn((Cin, N, kt) if kind == 4 else (N, Cin, kt), generator=g) / (Cin * kt) ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g)
    xd, wd_ = x.double(), w.double()
    if kind == 4:
        ref = F.conv_transpose1d(xd, wd_, b.double(), stride=2, padding=1)
    elif kind == 3:
        ref = F.conv1d(xd, wd_, b.double(), stride=2, padding=1)
    else:
        ref = F.conv1d(xd, wd_, b.double(), padding=kt // 2)
    Lout = ref.shape[2]
    res = torch.randn(B, N, Lout, generator=g).to(torch.bfloat16) if with_res else None
    if with_res:
        ref = ref + res.double()
    if slope >= 0:
        ref = F.leaky_relu(ref, slope)
    wp, Np, Kp = _pack(w.float(), kind)
    lda = Kp
    a_cl = torch.zeros(B * L, lda, dtype=torch.bfloat16)
    a_cl[:, :Cin] = x.permute(0, 2, 1).reshape(B * L, Cin)
    ldo = (N + 63) // 64 * 64
    r_cl = None
    if with_res:
        r_cl = torch.zeros(B * Lout, ldo, dtype=torch.bfloat16)
        r_cl[:, :N] = res.permute(0, 2, 1).reshape(B * Lout, N)
        r_cl = r_cl.cuda()
    a_cl, wp, bd = a_cl.cuda(), wp.cuda(), b.cuda()
    outs = {}
    try:
        for mode in (1, 0):
            _lib.check(lib.pt_set_option(b"conv8", mode))
            cb = torch.full((B * Lout, ldo), float("nan"), dtype=torch.bfloat16, device="cuda")
            cf = torch.full((B * Lout, ldo), float("nan"), device="cuda")
            _lib.check(lib.pt_conv1d_cl_bf16(_lib.ptr(a_cl), lda, _lib.ptr(wp), _lib.ptr(bd), N, Np, Kp, kind, B, L, _lib.ptr(cf), ldo,
                                             _lib.ptr(cb), ldo, 1 if slope >= 0 else 0, max(slope, 0.0), _lib.ptr(r_cl), ldo,
                                             _lib.stream()))
            torch.cuda.synchronize()
            outs[mode] = (cf.cpu(), cb.cpu())
    finally:
        _lib.check(lib.pt_set_option(b"conv8", 1))
    ref_rows = ref.permute(0, 2, 1).reshape(B * Lout, N)
    for mode in (1, 0):
        cf, cb = outs[mode]
        assert rel_err(cf[:, :N].double(), ref_rows) < 2e-5, mode                     # fp32 accumulation of exact bf16 products
        assert rel_err(cb[:, :N].double(), ref_rows) < 5e-3, mode                     # + one bf16 rounding of the output
        assert bool(torch.isnan(cf[:, N:]).all() | (cf[:, N:] == 0).all())            # pad columns: untouched or zeros, never garbage
    assert rel_err(outs[1][0][:, :N], outs[0][0][:, :N]) < 2e-5
    same = float((outs[1][1][:, :N] == outs[0][1][:, :N]).float().mean())
    assert same > 0.99, same          # bf16 outputs: identical except where the fp32 sums straddle a rounding boundary


def test_conv8_option_is_listed():
    import ctypes
    v = ctypes.c_int(-5)
    _lib.check(_lib.lib().pt_get_option(b"conv8", ctypes.byref(v)))
    assert v.value == 1


def test_pqvae_decoder_and_audio_encoder_bf16_through_conv8_match_oracle():
    """Module level at a size where the conv stacks run on the new kernel (>= 64 tiles of 256 x 256 per launch): PQ-VAE
    decode_codes and the audio ConditionEncoder in bf16 mode, conv8 on vs off, and against the fp32 oracle on sampled clips."""
    from oracle import probtalk_oracle as O
    from probtalk_amd.nets.modules import VQVAE_SC, ConditionEncoder
    from tests.helpers import l2_err, ref_state_dict
    lib = _lib.lib()
    vq = VQVAE_SC(376, 512, 128, 1024, 2, 512, q_type="pro", groups=4).eval()
    vq.load_state_dict(ref_state_dict("VQ"))
    vq.set_precision("bf16")
    ae = ConditionEncoder(768, 256, 2, 256).eval()
    ae.load_state_dict(ref_state_dict("AudEnc"))
    ae.set_precision("bf16")
    B, Tq = 448, 11
    g = torch.Generator().manual_seed(4)
    codes = torch.randint(0, 128, (B, Tq, 4), generator=g)
    aud = torch.randn(B, 768, 88, generator=g)
    outs = {}
    try:
        for mode in (1, 0):
            _lib.check(lib.pt_set_option(b"conv8", mode))
            outs[mode] = (vq.decode_codes(codes.cuda()).cpu(), ae(aud.cuda()).cpu())
    finally:
        _lib.check(lib.pt_set_option(b"conv8", 1))
    for k, what in ((0, "PQ-VAE decode_codes"), (1, "AudEnc")):
        e = l2_err(outs[1][k], outs[0][k])
        print(f"{what}: conv8 vs 4-wave kernel l2-rel {e:.2e}")
        # same bf16 products per layer (op-level test above: 2e-5), but every layer rounds its activations to bf16 from
        # slightly different fp32 sums and ~15 layers decorrelate: the two runs end as far apart as each is from the oracle
        assert e < 1e-2, what
    sel = [0, 1, 222, 447]
    sd = ref_state_dict("VQ")
    ref = O.vqvae_decode(sd, O.pq_quantize_all(sd, "vq_layer.", codes[sel]).transpose(1, 2))
    assert l2_err(outs[1][0][sel], ref) < 1e-2
    ref_a = O.condition_encoder(ref_state_dict("AudEnc"), aud[sel])
    assert l2_err(outs[1][1][sel], ref_a) < 1e-2


@pytest.mark.parametrize("kind,B,L,Cin,dim", [(2, 70, 300, 128, 128), (1, 130, 150, 128, 128), (3, 130, 300, 128, 256),
                                              (1, 260, 75, 256, 256), (3, 300, 150, 256, 512), (1, 470, 37, 512, 512)])
def test_conv8_gated_unit_matches_torch_and_the_4wave_kernel(kind, B, L, Cin, dim):
    """GatedConv.vert_stack + tanh(a) * sigmoid(b) (gated_pixelcnn_1d.py:74-83) with the weights interleaved in
    32-channel [a | b] blocks as runtime.hip packs them."""
    lib = _lib.lib()
    kt = [1, 3, 7, 4, 4][kind]
    g = torch.Generator().manual_seed(kind * 77 + L + dim)
    x = torch.randn(B, Cin, L, generator=g).to(torch.bfloat16)
    w = (torch.randn(2 * dim, Cin, kt, generator=g) / (Cin * kt) ** 0.5).to(torch.bfloat16)
    b = torch.randn(2 * dim, generator=g)
    h = F.conv1d(x.double(), w.double(), b.double(), stride=2 if kind == 3 else 1, padding=1 if kind == 3 else kt // 2)
    ref = torch.tanh(h[:, :dim]) * torch.sigmoid(h[:, dim:])
    Lout = ref.shape[2]
    # GEMM column 64 pr + c = a-channel 32 pr + c, 64 pr + 32 + c = b-channel dim + 32 pr + c
    perm = torch.cat([torch.cat([torch.arange(32 * pr, 32 * pr + 32), dim + torch.arange(32 * pr, 32 * pr + 32)])
                      for pr in range(dim // 32)])
    wp, Np, Kp = _pack(w.float()[perm], kind)
    a_cl = torch.zeros(B * L, Kp, dtype=torch.bfloat16)
    a_cl[:, :Cin] = x.permute(0, 2, 1).reshape(B * L, Cin)
    a_cl, wp, bd = a_cl.cuda(), wp.cuda(), b[perm].contiguous().cuda()
    ldo = (dim + 63) // 64 * 64
    outs = {}
    try:
        for mode in (1, 0):
            _lib.check(lib.pt_set_option(b"conv8", mode))
            cb = torch.full((B * Lout, ldo), float("nan"), dtype=torch.bfloat16, device="cuda")
            _lib.check(lib.pt_conv1d_cl_bf16(_lib.ptr(a_cl), Kp, _lib.ptr(wp), _lib.ptr(bd), 2 * dim, Np, Kp, kind, B, L, None, 0,
                                             _lib.ptr(cb), ldo, 2, 1.0, None, 0, _lib.stream()))
            torch.cuda.synchronize()
            outs[mode] = cb.cpu()
    finally:
        _lib.check(lib.pt_set_option(b"conv8", 1))
    ref_rows = ref.permute(0, 2, 1).reshape(B * Lout, dim)
    for mode in (1, 0):
        assert rel_err(outs[mode][:, :dim].double(), ref_rows) < 5e-3, mode
    assert float((outs[1][:, :dim] == outs[0][:, :dim]).float().mean()) > 0.99
