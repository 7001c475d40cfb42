"""The reference's own arithmetic path on THIS GPU: the oracle's torch modules (pinned to the reference by the golden
fixtures) moved to cuda, i.e. PyTorch-ROCm eager + MIOpen/rocBLAS fp32 -- what running the reference repo on an MI355X
amounts to.  Prints clips/s for the BASELINE configs[1] MoCo iteration.  A measurement tool (a baseline next to
bench.py's cpu_baseline leg): nothing in the product imports the oracle."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import moco as omoco, wrappers as owrap

def main(b=32, steps=5, bench=False):
    torch.backends.cudnn.benchmark = bench
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    model, ema = owrap.create_visual_model('R2P1D18', 16, 128, 'mlp', 'moco')
    model.to(dev).train(); ema.to(dev); omoco.set_key_encoder_mode(ema)
    contrast = omoco.RGBMoCo(128, K=4096, T=0.07).to(dev)
    crit = omoco.NCESoftmaxLoss()
    opt = omoco.make_optimizer(model, 0.06, 0.9, 5e-4)
    x = torch.randn(b, 6, 16, 112, 112, device=dev)
    t0 = time.time()
    for i in range(2):
        omoco.moco_train_step(model, ema, contrast, crit, opt, x, 0.999)
    torch.cuda.synchronize(); print('warm-up %.1fs' % (time.time() - t0), flush=True)
    t0 = time.time()
    for i in range(steps):
        omoco.moco_train_step(model, ema, contrast, crit, opt, x, 0.999)
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps
    print('bench=%s  %.1f ms/step  %.1f clips/s' % (bench, dt * 1e3, b / dt), flush=True)

if __name__ == '__main__':
    main(bench='--bench' in sys.argv)
