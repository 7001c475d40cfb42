"""InfoNCE forward timing the way bench.py reports it (20 back-to-back repetitions inside one hipGraph), after a
one-second MFMA burn so the clocks are where a training step leaves them (these kernels are short enough to be
timed at idle clocks otherwise: 2x run-to-run spread)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('video-graph-ssl_amd')
a = torch.randn(8192, 8192, device='cuda')
t = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
t.record()
for _ in range(40): a @ a
t2.record(); torch.cuda.synchronize()
import os
for rep in range(3):
    print(bench.infonce_timing(pkg, 32, warm=os.environ.get('NCE_WARM', '0') == '1'), flush=True)
