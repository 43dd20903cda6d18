"""cProfile of the eager train() host path at a tiny batch (GPU time negligible -> wall = host overhead)."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import bench
dev = torch.device("cuda:0")
pol, src, tar, cfg = bench.build(dev, 0, 128, 0)
for _ in range(20):
    pol.train(src, tar, 128, None, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    pol.train(src, tar, 128, None, None)
torch.cuda.synchronize()
print("eager host-bound ms/step", (time.perf_counter() - t0) / 300 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    pol.train(src, tar, 128, None, None)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
