"""pt_pq_quantize_static: the no-allocation variant of pt_pq_quantize (staging images in a file-scope device array
instead of a caller workspace).  Checked through the C ABI against pt_pq_quantize itself (bit-exact: integer indices and
gathered fp32 code rows) and against the CPU oracle, on two calls in a row, the second with a smaller input (the array
keeps the first call's contents).  Collected last (file name) so a failure here hides no other test."""
import pytest
import torch

from oracle import probtalk_oracle as O
from probtalk_amd.weights import make_input
from tests.helpers import ref_state_dict

pytestmark = pytest.mark.gpu


def _both(vq, z):
    from probtalk_amd import _lib
    L = _lib.lib()
    h = vq._handle()
    B, C, W = z.shape
    G = 4
    zd = z.cuda().contiguous()
    ws = torch.empty(2 * B * W * C * 4 + 1024, dtype=torch.uint8, device="cuda")
    q0 = torch.full((B, C, W), float("nan"), device="cuda")
    i0 = torch.full((B, W, G), -1, dtype=torch.int64, device="cuda")
    _lib.check(L.pt_pq_quantize(h.h, _lib.ptr(zd), B, W, _lib.ptr(q0), _lib.ptr(i0), _lib.ptr(ws), ws.numel(), _lib.stream()))
    q1 = torch.full((B, C, W), float("nan"), device="cuda")
    i1 = torch.full((B, W, G), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    # the device's free bytes (hipMemGetInfo) see a hipMalloc made inside the library; torch.cuda.memory_allocated()
    # only sees torch's own caching allocator.  pt_pq_quantize above was the warm-up (code object, kernel attribute).
    before = torch.cuda.memory_allocated()
    free_before = torch.cuda.mem_get_info()[0]
    _lib.check(L.pt_pq_quantize_static(h.h, _lib.ptr(zd), B, W, _lib.ptr(q1), _lib.ptr(i1), _lib.stream()))
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before, "device memory was allocated inside pt_pq_quantize_static"
    # and under stream capture, where a hipMalloc or a synchronous call would make the capture fail
    q2 = torch.full((B, C, W), float("nan"), device="cuda")
    i2 = torch.full((B, W, G), -1, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.capture_begin()
        rc = L.pt_pq_quantize_static(h.h, _lib.ptr(zd), B, W, _lib.ptr(q2), _lib.ptr(i2), _lib.stream())
        graph.capture_end()
    assert rc == 0, L.pt_last_error()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(q2, q1) and torch.equal(i2, i1), "graph replay of pt_pq_quantize_static differs from the eager call"
    return (q0.cpu(), i0.cpu()), (q1.cpu(), i1.cpu())


def test_pq_quantize_static_matches_workspace_entry_point():
    from probtalk_amd import _lib
    from probtalk_amd.nets.modules import VQVAE_SC
    vq = VQVAE_SC(376, 512, 128, 1024, 2, 512, q_type="pro", groups=4).eval()
    sd = ref_state_dict("VQ")
    vq.load_state_dict(sd)
    # two calls in a row on the same GPU; the second is smaller, so stale rows of the first stay behind it in the array
    for tag, shape in (("na.z1", (64, 512, 37)), ("na.z2", (5, 512, 11))):
        z = make_input(tag, shape)
        (q0, i0), (q1, i1) = _both(vq, z)
        assert torch.equal(i1, i0), "indices differ from pt_pq_quantize"
        assert torch.equal(q1, q0), "quantized latents differ from pt_pq_quantize"
        q_ref, idx_ref = O.pq_quantize(sd, "vq_layer.", z, 4)
        mism = i1 != idx_ref
        if int(mism.sum()):
            marg = O.pq_margins(sd, "vq_layer.", z, 4)
            assert float(marg[mism].max()) < 1e-5 * float((z ** 2).sum(1).mean()), "index mismatches are not near-ties"
        same = (i1 == idx_ref).all(-1)
        assert torch.equal(q1.permute(0, 2, 1)[same], q_ref.permute(0, 2, 1)[same])
    # beyond the static maximum: a clear error, nothing launched
    L = _lib.lib()
    h = vq._handle()
    z = torch.zeros(4097, 512, 1, device="cuda")
    q = torch.empty_like(z)
    idx = torch.empty(4097, 1, 4, dtype=torch.int64, device="cuda")
    rc = L.pt_pq_quantize_static(h.h, _lib.ptr(z), 4097, 1, _lib.ptr(q), _lib.ptr(idx), _lib.stream())
    assert rc != 0 and b"exceed the static scratch" in L.pt_last_error()
