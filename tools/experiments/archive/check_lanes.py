"""Encoder lanes (ResNet.split_lanes) A/B: outputs of the multi-stream trunk against the single-stream one, repeated."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fgvc_amd.mmpt_api as api
from fgvc_amd.mmpt_api.backbones import ResNet
dev = torch.device("cuda")
torch.manual_seed(0)
net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none"))
net.init_weights()
net = net.to(dev).eval()
for shape in ((3, 3, 76, 132), (8, 3, 480, 854)):
    x = torch.randn(*shape, device=dev)
    with torch.no_grad():
        ResNet.split_lanes = 1
        ref = net(x).clone()
        ref2 = net(x).clone()
        print(shape, "lanes 1 twice:", float((ref - ref2).abs().max()), "scale", float(ref.abs().max()))
        for lanes in (2, 3):
            ResNet.split_lanes = lanes
            for rep in range(4):
                y = net(x).clone()
                torch.cuda.synchronize()
                print(shape, "lanes", lanes, "rep", rep, "max diff vs lanes 1:", float((y - ref).abs().max()))
