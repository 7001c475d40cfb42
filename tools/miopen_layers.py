"""Per-layer timing of ATen/MIOpen fp32 conv3d (cudnn.benchmark=True) on the R(2+1)D-18 layer set of
tools/conv_micro.py -- what the reference's GPU path spends per layer, next to ours."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('GCA_AUTOTUNE', '0')
torch.backends.cudnn.benchmark = True
from conv_micro import LAYERS
dev = torch.device('cuda:0')
def ev(fn, reps=5):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name in (sys.argv[1].split(',') if len(sys.argv) > 1 else LAYERS):
    C, D, H, W, K, k, s, p = LAYERS[name]
    x = torch.randn(32, C, D, H, W, device=dev, requires_grad=True)
    w = (torch.randn((K, C) + k, device=dev) * 0.05).requires_grad_(True)
    y = F.conv3d(x, w, None, s, p)
    fl = 2.0 * y.numel() * C * k[0] * k[1] * k[2]
    dy = torch.randn_like(y)
    t0 = time.time()
    tf = ev(lambda: F.conv3d(x, w, None, s, p))
    td = ev(lambda: torch.autograd.grad(F.conv3d(x, w.detach(), None, s, p), x, dy)) - tf
    tw = ev(lambda: torch.autograd.grad(F.conv3d(x.detach(), w, None, s, p), w, dy)) - tf
    print('%s  GF %7.2f | fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF | (search+run %.0fs)'
          % (name, fl / 1e9, tf, fl / 1e9 / tf, td, fl / 1e9 / max(td, 1e-3), tw, fl / 1e9 / max(tw, 1e-3), time.time() - t0), flush=True)
