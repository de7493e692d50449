"""GPU probe: error of the DLA-34 network outputs against the golden outputs recorded from the reference's own
modules (tests/golden/net_dla34.npz), per head, as a fraction of the head's max-norm, for the inference path
(prepare_inference).  `python tools/probe_net_error.py exact_f32` for the exact-fp32 arithmetic."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "golden"))
import numpy as np, torch
import cases
from centerpoly_amd.models.model import create_model
from centerpoly_amd import arithmetic
arithmetic.configure(sys.argv[1] if len(sys.argv) > 1 else "split_bf16")

gold = dict(np.load(os.path.join(root, "tests", "golden", "net_dla34.npz"), allow_pickle=True))
shapes = {k: tuple(v) for k, v in json.loads(str(gold["shapes"])).items()}
m = create_model("dla_34", dict(cases.HEADS), 256)
m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in cases.fill_weights(shapes).items()})
m = m.cuda().eval()
x = torch.from_numpy(cases.net_input("dla")).cuda()
for mode in ("plain", "prepare_inference"):
    if mode == "prepare_inference":
        m.prepare_inference()
    with torch.no_grad():
        out = m(x)[0]
    print(mode, {h: "%.2e" % (np.abs(out[h].cpu().numpy() - gold["s0_" + h]).max() / np.abs(gold["s0_" + h]).max())
                 for h in dict(cases.HEADS)})
