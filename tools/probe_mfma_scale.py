"""Layout experiment for v_mfma_scale_f32_16x16x128_f8f6f4 (tools build): which (lane, byte) positions of the A/B
registers multiply each other, and which lane's scale byte applies to which position."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_vlb_amd._lib import lib

lib.vlb_mfma_scale_probe.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
ONE = 0x38       # 1.0 in e4m3


def run(a_bytes, b_bytes, sa, sb, mode=0):
    a = a_bytes.to(dev).contiguous().view(torch.int32); b = b_bytes.to(dev).contiguous().view(torch.int32)
    d = torch.zeros(64, 4, device=dev)
    lib.vlb_mfma_scale_probe(a.data_ptr(), b.data_ptr(), sa.to(dev).data_ptr(), sb.to(dev).data_ptr(), d.data_ptr(), mode,
                             torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d.cpu()


ones = torch.full((64, 32), ONE, dtype=torch.uint8)
s1 = torch.full((64,), 127, dtype=torch.int32)
print("all ones, unit scales -> every D should be 128:", run(ones, ones, s1, s1).unique().tolist())
# which A position pairs with which B position: A nonzero only at (lane la, byte ja); B nonzero at (lb, jb)
a = torch.zeros(64, 32, dtype=torch.uint8); a[5 + 16 * 2, 7] = ONE          # row 5, lane group 2, byte 7
hits = []
for g in range(4):
    for j in range(32):
        b = torch.zeros(64, 32, dtype=torch.uint8); b[9 + 16 * g, j] = ONE     # col 9
        d = run(a, b, s1, s1)
        if d.abs().sum() > 0:
            nz = d.nonzero()
            hits.append((g, j, [(int(l), int(r), float(d[l, r])) for l, r in nz]))
print("A(row5,g2,byte7) x B(col9,g,byte j) nonzero for:", hits)
# scale association: A data at (row 3, group g0, byte j0) = 1, B all ones; scale_a of lane group g = 2^g (same for all rows)
sg = torch.tensor([127 + (l // 16) for l in range(64)], dtype=torch.int32)
for g0 in range(4):
    row = []
    for j0 in (0, 8, 15, 16, 24, 31):
        a = torch.zeros(64, 32, dtype=torch.uint8); a[3 + 16 * g0, j0] = ONE
        d = run(a, ones, sg, s1)
        row.append(float(d.abs().max()))
    print(f"A data in lane group {g0}, bytes (0,8,15,16,24,31): value picked up = {row}  (2^g of the scale that applied)")
# does the scale come from the lane that holds the data, or from a fixed set of lanes?  scale_a = 2 only in ONE lane
for sl in (3, 19, 35, 51, 4):
    sa = s1.clone(); sa[sl] = 128
    a = torch.zeros(64, 32, dtype=torch.uint8); a[3::16, :] = ONE            # row 3, all four lane groups, all bytes
    d = run(a, ones, sa, s1)
    print(f"scale 2 in lane {sl}: D(row 3) = {float(d.abs().max())}   (128 = no effect, 160 = one 32-block doubled)")
# byte selection inside the scale register (opsel 0): put the scale in byte 1 instead
sa = torch.full((64,), 127 | (130 << 8), dtype=torch.int32)
print("scale byte1=130, byte0=127:", run(ones, ones, sa, s1).unique().tolist())
def dmat(d):
    """D as a [16 rows, 16 cols] matrix: col = lane & 15, row = (lane >> 4) * 4 + reg."""
    m = torch.zeros(16, 16)
    for l in range(64):
        for r in range(4):
            m[(l >> 4) * 4 + r, l & 15] = d[l, r]
    return m


print("---- byte selection (all-ones data, uniform words): D = 128 * scaleA * scaleB")
def w(b0, b1=127, b2=127, b3=127):
    v = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24)
    return torch.full((64,), v - (1 << 32) if v >= (1 << 31) else v, dtype=torch.int32)
for mode, name in ((0, "opsel a=0 b=0"), (1, "opsel a=0 b=1"), (2, "opsel a=1 b=0"), (3, "opsel a=2 b=3")):
    for label, va, vb in (("v_a=[130,..] v_b=127", w(130), w(127)), ("v_a=127 v_b=[130,127,..]", w(127), w(130)),
                          ("v_a=127 v_b=[127,131,..]", w(127), w(127, 131)), ("v_a=[127,131] v_b=127", w(127, 131), w(127)),
                          ("v_a=127 v_b=[127,127,129,132]", w(127), w(127, 127, 129, 132)), ("v_a=[127,127,129,132] v_b=127", w(127, 127, 129, 132), w(127))):
        print(f"{name}: {label}: D/128 = {[x / 128 for x in run(ones, ones, va, vb, mode).unique().tolist()]}")
print("---- scale <-> data association (mode 1: A scale = byte0, B scale = byte1 of the scale_b register)")
sg = torch.tensor([(127 + (l // 16)) | (127 << 8) for l in range(64)], dtype=torch.int32)
for g0 in range(4):
    row = []
    for j0 in (0, 8, 15, 16, 24, 31):
        a = torch.zeros(64, 32, dtype=torch.uint8); a[3 + 16 * g0, j0] = ONE
        d = run(a, ones, s1, sg, 1)
        row.append(float(d.abs().max()))
    print(f"A data in lane group {g0}, bytes (0,8,15,16,24,31): picked-up A scale 2^g with g = {[int(torch.tensor(v).log2()) for v in row]}")
sgb = torch.tensor([127 | ((127 + (l // 16)) << 8) for l in range(64)], dtype=torch.int32)
for g0 in range(4):
    row = []
    for j0 in (0, 8, 15, 16, 24, 31):
        b = torch.zeros(64, 32, dtype=torch.uint8); b[3 + 16 * g0, j0] = ONE
        d = run(ones, b, s1, sgb, 1)
        row.append(float(d.abs().max()))
    print(f"B data in lane group {g0}, bytes (0,8,15,16,24,31): picked-up B scale 2^g with g = {[int(torch.tensor(v).log2()) for v in row]}")
# row association: scale 2^(row) in lanes of group 0 only
sr = torch.tensor([(127 + (l % 16)) | (127 << 8) for l in range(64)], dtype=torch.int32)
a = torch.zeros(64, 32, dtype=torch.uint8); a[:16, 0] = ONE        # every row, group 0, byte 0
d = dmat(run(a, ones, s1, sr, 1))
print("A rows pick up scale 2^r from lane r of group 0:", [int(x) for x in d[:, 0].log2().tolist()])
