import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
import test_graph_gpu as T
from multimodal_amd import harness
dev = torch.device("cuda:0")
sched = [1e-2, 5e-3, 2e-3]
base = dict(harness.CONFIGS["tiny"])
for tag, over in (("default", {}), ("unfused_head", {"fused_head_loss": False}), ("no_prefix", {"prefix_len": 0})):
    harness.CONFIGS["tiny"] = dict(base, **over)
    l1, p1, m1 = T._run(dev, "tiny", False, 3, sched)
    l2, p2, m2 = T._run(dev, "tiny", False, 3, sched)
    l3, p3, m3 = T._run(dev, "tiny", True, 3, sched)
    l4, p4, m4 = T._run(dev, "tiny", True, 3, sched)
    print(tag, "E", l1, "E", l2, "G", l3, "G", l4)
