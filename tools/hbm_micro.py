"""The two HBM-bound head kernels of the path as bench.py times them (distinct buffer sets > 256 MB per lap, one hipGraph):
InfoNCE forward at K = 4096 / 65536 and the temporal-graph message passing at the configs[3] site.  Run under
tools/pmc_hbm.sh for the FETCH_SIZE / WRITE_SIZE passes."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('video-graph-ssl_amd')
a = torch.randn(8192, 8192, device='cuda')
for _ in range(20):
    a @ a
torch.cuda.synchronize()
print(bench.infonce_timing(pkg, 32), flush=True)
print(bench.graph_timing(pkg), flush=True)
