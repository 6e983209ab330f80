"""First kernel of the forward whose output differs between a large batch and the same crops in a small batch
(different band configurations), and where: python scripts/diag/nan_probe.py [big batch] [small batch]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from face_recognition_amd import synthetic
from face_recognition_amd.face_model import MobileFaceNet
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
b = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sd = synthetic.make_state_dict(1, 512)
net = MobileFaceNet(512); net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net = net.to(dev).eval()
x = torch.from_numpy(synthetic.make_crops(B, 13)).to(dev)
net(x)
names = [r[0] for r in net.op_table()]
for n in range(1, len(names)):
    big = net.debug_prefix(x, n)[:b].cpu()
    small = net.debug_prefix(x[:b].contiguous(), n).cpu()
    bad = ~torch.isfinite(big)
    diff = (big - small).abs()
    diff[bad] = 1e9
    print("%2d %-18s finite %s  max diff %.3g" % (n, names[n - 1], bool((~bad).all()), diff.max().item()))
    if diff.max() > 0:
        img, pos = np.unravel_index(int(diff.argmax()), diff.shape)
        w = diff[img].nonzero().flatten()
        print("   image", img, "first/last differing offsets", w[0].item(), w[-1].item(), "count", len(w), "of", diff.shape[1])
        v = big[img].view(-1, 64) if names[n - 1].startswith("conv_3") else None
        if v is not None:
            s_ = small[img].view(-1, 64)
            for px in (12 * 28 + 15, 12 * 28 + 16, 12 * 28 + 17, 12 * 28 + 27, 13 * 28 + 16, 13 * 28 + 27):
                print("   px row %d col %d: big %s | small %s" % (px // 28, px % 28, v[px, :6].tolist(), s_[px, :6].tolist()))
            nf = (~torch.isfinite(v)).any(1).nonzero().flatten().tolist()
            print("   non-finite pixels (row, col):", [(p // 28, p % 28) for p in nf][:60])
        break
