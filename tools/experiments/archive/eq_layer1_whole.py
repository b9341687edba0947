import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fgvc_amd.mmpt_api.backbones import ResNet
dev = torch.device("cuda:0")
model = bench.build_tracker(bench.WORKLOADS["cfg2_480p_8f"], dev)
x = torch.randn(8, 3, 480, 854, device=dev)
a = model.get_feats_hwc(x, split=True)[0].clone()
ResNet.layer1_whole_batch = False
b = model.get_feats_hwc(x, split=True)[0].clone()
ResNet.layer1_whole_batch = True
c = model.get_feats_hwc(x, split=True)[0]
print("whole == lanes:", torch.equal(a, b), torch.equal(a, c), "overflow", model.backbone.check_overflow())
