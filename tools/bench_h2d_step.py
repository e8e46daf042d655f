"""PCIe-inclusive step rate: batches start in pinned HOST memory and reach the device through
DevicePrefetcher (side-stream copy of batch i+1 under step i), versus the same batches resident in HBM."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd.datamodule import DevicePrefetcher
from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig
from phantom_vlb_amd.synthetic import synthetic_batch

lora = "--frozen" not in sys.argv
B = 3 if lora else 5
cfg = VLBLitModuleConfig(model_path="none", freeze_backbone=not lora, use_lora=lora, lora_r=16 if lora else None,
                         lora_alpha=32 if lora else None, lora_dropout=0.1 if lora else None, dropout_rate=0.1, num_target=2048,
                         l2_lambda=1e-3, lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-2,
                         lr_scheduler_name="CosineAnnealingLR", last_epoch=-1, t_max=50000, geometry="7b")
warnings.simplefilter("ignore")
m = VLBLitModule(cfg); m.configure_model()
opt, sch = m.configure_optimizers(); opt, sch = opt[0], sch[0]["scheduler"]
dev = m.device
host = [synthetic_batch(m.geometry, B, seed=100 + i) for i in range(4)]
host = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()} for b in host]


class Loop:                      # a "DataLoader" that cycles the pinned batches
    sampler = None
    def __init__(self, n): self.n = n
    def __len__(self): return self.n
    def __iter__(self):
        for i in range(self.n):
            yield host[i % len(host)]


def run(loader, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for batch in loader:
        m.training_step(batch); opt.step(); sch.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

run(DevicePrefetcher(Loop(3), dev), 3)
n = 12
t_pre = run(DevicePrefetcher(Loop(n), dev), n)
t_sync = run(Loop(n), n)                      # the step's own .to(device) calls (synchronous copies from pinned memory)
res = [{k: (v.to(dev) if k not in ("language", "padvals") else v) for k, v in b.items()} for b in host]
class Res(Loop):
    def __iter__(self):
        for i in range(self.n): yield res[i % len(res)]
t_res = run(Res(n), n)
mb = sum(v.numel() * v.element_size() for v in host[0].values() if torch.is_tensor(v)) / 1e6
print(f"{'lora' if lora else 'frozen'} B={B}: batch {mb:.1f} MB | resident {t_res*1e3:.1f} ms/step = {B/t_res:.2f} clips/s | "
      f"prefetched from pinned host {t_pre*1e3:.1f} ms = {B/t_pre:.2f} clips/s | in-step copies {t_sync*1e3:.1f} ms = {B/t_sync:.2f} clips/s")
