"""GPU probe: which convolutions of an inference pass still go to the library (F.conv2d calls with their shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from centerpoly_amd import synth

arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
model, _ = bench.build_model(torch.device("cuda"), train=False, arch=arch)
x = torch.from_numpy(synth.normal("bench/input", (1, 3, 1024, 2048))).cuda()
orig = F.conv2d
calls = []
def logged(inp, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    calls.append((tuple(inp.shape), tuple(w.shape), stride, padding, dilation, groups))
    return orig(inp, w, b, stride, padding, dilation, groups)
F.conv2d = logged
torch.nn.functional.conv2d = logged
with torch.no_grad():
    model(x)
F.conv2d = orig
for c in calls:
    print(c)
print(len(calls), "library convolutions")
