import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
M, N, K = 256, 256, 128
def go(a, w, label):
    aq, sa = ops.quantize_mxfp8(a.to(BF).to(dev)); wq, sw = ops.quantize_mxfp8(w.to(BF).to(dev))
    out = ops.gemm_mxfp8(aq, sa, wq, sw).float().cpu()
    ref = a.to(BF).float() @ w.to(BF).float().t()
    bad = (out - ref).abs() > 0.02 * ref.abs().max()
    print(label, "bad:", int(bad.sum()), "of", bad.numel(), "| bad rows:", bad.any(1).nonzero().flatten()[:20].tolist(), "| bad cols:", bad.any(0).nonzero().flatten()[:20].tolist())
    if bad.any():
        i, j = bad.nonzero()[0].tolist(); print("   first bad", i, j, float(out[i, j]), float(ref[i, j]), " scales sa", sa[i].tolist(), "sw", sw[j].tolist())
go(torch.ones(M, K), torch.ones(N, K), "ones x ones")
go(torch.arange(M).float()[:, None].remainder(7).add(1).expand(M, K).contiguous(), torch.ones(N, K), "row-const A (1..7) x ones")
go(torch.ones(M, K), torch.arange(N).float()[:, None].remainder(5).add(1).expand(N, K).contiguous(), "ones x row-const W")
a = torch.ones(M, K); a[:, 32:64] = 2; a[:, 64:96] = 4; a[:, 96:] = 8
go(a, torch.ones(N, K), "A kblocks 1,2,4,8 x ones")
a = torch.ones(M, K); a[:, 32:64] = 16; 
go(a * torch.arange(M).float()[:, None].remainder(3).add(1), torch.ones(N, K), "A scaled blocks x ones")
a = torch.zeros(M, K); a[:, 5] = 1; w = torch.zeros(N, K); w[:, 5] = 1
go(a, w, "single k=5")
a = torch.zeros(M, K); a[:, 37] = 3; w = torch.zeros(N, K); w[:, 37] = 1
go(a, w, "single k=37")
torch.manual_seed(0)
go(torch.randn(M, K), torch.randn(N, K), "randn x randn")
go(torch.randn(M, K) * torch.rand(M, 1).mul(4).exp(), torch.randn(N, K) * 0.05, "test-like")
print("---- asymmetric")
a = torch.zeros(M, K); a[:, 37] = 3
go(a, torch.ones(N, K), "A single k=37 x W ones")
w = torch.zeros(N, K); w[:, 37] = 3
go(torch.ones(M, K), w, "A ones x W single k=37")
w = torch.ones(N, K); w[:, 32:64] = 2; w[:, 64:96] = 4; w[:, 96:] = 8
go(torch.ones(M, K), w, "ones x W kblocks 1,2,4,8")
a = torch.ones(M, K) * 1e-3; a[:, 37] = 3
go(a, torch.ones(N, K), "A tiny + k=37 x ones")
w = torch.ones(N, K) * 1e-3; w[:, 37] = 3
go(torch.ones(M, K), w, "ones x W tiny + k=37")
