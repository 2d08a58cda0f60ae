import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist, torch.multiprocessing as mp


def w(rank, port):
    faulthandler.dump_traceback_later(60, exit=True)
    import time
    t00 = time.time()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from probtalk_amd import synthetic, dist as D
    from probtalk_amd.weights import make_input
    import threading
    D.GLOO_DEVICE_TENSORS = len(sys.argv) > 2 and sys.argv[2] == 'device'
    orig = D.GradientReducer._launch
    def dbg(self, i):
        live = [p for p in self.buckets[i] if p.grad is not None]
        print(f"{time.time()-t00:6.1f}s r{rank} launch bucket {i} live {len(live)}/{len(self.buckets[i])} n={sum(p.numel() for p in live)} thread={threading.current_thread().name}", flush=True)
        return orig(self, i)
    D.GradientReducer._launch = dbg
    pred, _ = synthetic.build_pipeline(seed=0)
    pred.train_dropout = 0.0
    B, T_ = 2, 88
    bat = {"aud_feat": make_input(f"dt.aud{rank}", (B, 768, T_)), "poses": 0.3 * make_input(f"dt.poses{rank}", (B, 330, T_)),
           "expression": 0.3 * make_input(f"dt.expr{rank}", (B, 100, T_)), "speaker": torch.tensor([rank, 3 - rank]),
           "epoch": 10, "_mask": torch.zeros(B, 1, T_), "_keep_scores": torch.rand(B, 44, generator=torch.Generator().manual_seed(rank))}
    for step in range(2):
        _, ld = pred(bat)
        print(f"{time.time()-t00:6.1f}s", rank, "step", step, ld["loss"], pred._reducer.launched_in_backward, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(w, args=(int(sys.argv[1]),), nprocs=2, join=True)
