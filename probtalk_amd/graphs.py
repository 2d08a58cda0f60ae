"""HIP-graph replay of the full inference step for small batches.

Up to about 64 clips the step is launch-bound: one pass of the path enqueues ~1,900 kernels (8 MaskGIT passes x 6 decoder
layers, the conv stacks, the refiner) and the host needs ~8 us per launch, 15 ms per step whatever the GPU does.  The
library calls allocate nothing, synchronise nothing and take every run-time value (including the sampler's seed, see
`pt_predictor_sample_seed_dev`) from device memory, so the whole step can be captured once per (B, T) and replayed.
The reference has no counterpart (eager PyTorch)."""
import torch

from .synthetic import holistic_step


class GraphedHolisticStep:
    """predictor.infer_on_batch -> refiner.infer_on_batch (ai1.py:51-57) for a fixed (B, T), captured into one HIP graph.

    __call__(aud, gt_poses, mask, ids, seed=None) copies the inputs into the graph's static buffers, writes the seed and
    replays; the returned (B, 376, T) tensor is the graph's static output buffer (clone it to keep it past the next call).
    Weights are read in place: call `recapture()` after loading new weights or switching precision."""

    def __init__(self, pred, ref, B, T, warmup=2):
        self.pred, self.ref, self.B, self.T = pred, ref, B, T
        dev = "cuda"
        self.aud = torch.zeros(B, 768, T, device=dev)
        self.gt = torch.zeros(B, 376, T, device=dev)
        self.mask = torch.zeros(B, 1, T, device=dev)
        self.ids = torch.zeros(B, dtype=torch.int64, device=dev)
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self._warmup = warmup
        self.recapture()

    def _run(self):
        return holistic_step(self.pred, self.ref, self.aud, self.gt, self.mask, self.ids, sync=False, seed=self.seed)

    def recapture(self):
        # warm-up on a side stream (torch's capture rule): weight packing, workspace growth and one-time kernel attribute
        # calls all happen here, none inside the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self._warmup):
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._run()
        # the library's per-model workspaces were sized by the warm-up and are written by the graph: keep them alive
        self._keep = [m._handle()._ws for m in (self.pred.AudEnc, self.pred.Predictor, self.pred.VQ, self.ref.RTransformer)]

    def __call__(self, aud, gt_poses, mask, ids, seed=None):
        self.aud.copy_(aud.expand_as(self.aud))
        self.gt.copy_(gt_poses.expand_as(self.gt))
        self.mask.copy_(mask.expand_as(self.mask))
        self.ids.copy_(torch.as_tensor(ids).reshape(-1).expand_as(self.ids))
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.seed.fill_(int(seed))
        self.graph.replay()
        return self.out
