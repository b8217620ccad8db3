import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.conv1d_roofline_run(torch.device('cuda', 0), iters=10)
print(r['ms_per_iter'], r['roofline']['frac'])
