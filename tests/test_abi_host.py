"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares,
host mirror builds the reference's checkpoint key grammar, and ops refuse host tensors."""
import json
import os
import re

import numpy as np
import pytest
import torch

import cases
from centerpoly_amd import _C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "centerpoly_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = _header_symbols()
    assert len(syms) >= 16
    L = _C.lib()                       # raises NativeLibraryMissing if not built
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert set(syms) == set(_C.EXPORTS)
    assert L.cp_abi_version() == _C.ABI_VERSION == 3
    assert L.cp_build_arch() == b"gfx950"
    assert L.cp_strerror(-2) == b"unsupported shape or option"


def test_graft_entry_build_and_header_version_agree():
    """__graft_entry__.build() (the driver's "does it build" check) runs here, and the binding's ABI_VERSION is the
    header's CP_ABI_VERSION (round 3 bumped it to 2: build() must follow)."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__
    __graft_entry__.build()
    with open(os.path.join(root, "include", "centerpoly_hip.h")) as fh:
        m = re.search(r"#define\s+CP_ABI_VERSION\s+(\d+)", fh.read())
    assert m and int(m.group(1)) == _C.ABI_VERSION == _C.lib().cp_abi_version()


def test_argument_validation_without_gpu():
    """Entry points validate before touching the device: callable with no GPU."""
    L = _C.lib()
    # 256 tiles x K keys of 8 bytes + the merge level's 8 lists of K keys
    assert L.cp_polydet_decode_workspace_bytes(1, 8, 256, 512, 128) == (256 + 8) * 128 * 8
    assert L.cp_polydet_decode(None, None, None, None, 1, 8, 4, 4, 32, 8, 0, None, None, None,
                               None, 0, None) == -1
    assert L.cp_sigmoid_focal_forward(None, None, 16, None, None, None, 0, None) == -1
    s = _C.DcnShape(1, 8, 4, 4, 8, 3, 3, 1, 1, 1, 1)
    assert L.cp_dcn_v2_forward(s, None, None, 0, None, 0, 0, None, None, None, None, 0, 0, None,
                               None, 0, None) == -1
    # small-spatial, wide layers split K and need a workspace; big maps do not
    assert L.cp_dcn_v2_forward_workspace_bytes(_C.DcnShape(1, 512, 32, 64, 256, 3, 3, 1, 1, 1, 1)) > 0
    # no K split at this shape: only the permuted weights of the split-bf16 contraction
    # (8 sixteen-row tiles x 8 chunks of 8 channels x 3 k-steps x hi/lo x 64 lanes x 16 B)
    assert L.cp_dcn_v2_forward_workspace_bytes(_C.DcnShape(1, 64, 256, 512, 64, 3, 3, 1, 1, 1, 1)) == 8 * 8 * 3 * 2 * 64 * 16
    # round 4: split activations and the base pair validate on the host too
    assert L.cp_dla_base_pair_supported(1024, 2048) == 1 and L.cp_dla_base_pair_supported(37, 100) == 1
    assert L.cp_dla_base_pair_supported(64, 30) == 0 and L.cp_dla_base_pair_supported(0, 64) == 0
    assert L.cp_dla_base_pair_supported(16384, 16384) == 0                       # (32-bit offsets within an image)
    assert L.cp_dla_base_pair_forward(None, None, None, None, None, None, 1, 64, 64, None) == -1
    assert L.cp_activation_split(None, None, 1, 16, 8, 8, None) == -1
    assert L.cp_conv_mfma_forward_split(None, 1, None, None, None, None, 0, 1, 64, 8, 32, 64, 9, 1, 0, None) == -1


def test_ops_refuse_host_tensors():
    from centerpoly_amd.models.decode import polydet_decode
    from centerpoly_amd.models.losses import RegL1Loss
    heat = torch.rand(1, 2, 8, 8)
    with pytest.raises(_C.NativeError):
        polydet_decode(heat, torch.zeros(1, 4, 8, 8), torch.zeros(1, 1, 8, 8), K=4)
    with pytest.raises(_C.NativeError):
        RegL1Loss()(torch.zeros(1, 2, 8, 8), torch.ones(1, 3, dtype=torch.uint8),
                    torch.zeros(1, 3, dtype=torch.int64), torch.zeros(1, 3, 2))


def test_detector_rejects_cpu():
    from centerpoly_amd.detectors.detector_factory import detector_factory
    from centerpoly_amd.opts import opts
    opt = opts().init(["polydet", "--gpus", "-1"])
    with pytest.raises(RuntimeError):
        detector_factory["polydet"](opt)


@pytest.mark.parametrize("arch,gold", [("dla_34", "net_dla34"), ("smallhourglass", "net_hourglass1"),
                                       ("hourglass", "net_hourglass2")])
def test_checkpoint_key_grammar(arch, gold, golden):
    from centerpoly_amd.models.model import create_model
    shapes = {k: tuple(v) for k, v in json.loads(str(golden(gold)["shapes"])).items()}
    m = create_model(arch, dict(cases.HEADS), 256 if "dla" in arch else 64)
    own = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert own == shapes
    assert list(own) == list(shapes)           # same registration order too


def test_model_save_load_roundtrip(tmp_path):
    from centerpoly_amd.models.model import create_model, load_model, save_model
    m = create_model("dla_34", dict(cases.HEADS), 256)
    opt_ = torch.optim.Adam(m.parameters(), 4e-6)
    p = str(tmp_path / "model_last.pth")
    save_model(p, 7, m, opt_)
    ck = torch.load(p)
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and ck["epoch"] == 7
    m2 = create_model("dla_34", dict(cases.HEADS), 256)
    m2, o2, ep = load_model(m2, p, torch.optim.Adam(m2.parameters(), 4e-6), resume=True, lr=4e-6,
                            lr_step=[5, 90])
    assert ep == 7 and abs(o2.param_groups[0]["lr"] - 4e-7) < 1e-12
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_dcn_init_is_zero_offset():
    from centerpoly_amd.models.networks.DCNv2.dcn_v2 import DCN
    d = DCN(8, 4, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    assert set(dict(d.named_parameters())) == {"weight", "bias", "conv_offset_mask.weight",
                                               "conv_offset_mask.bias"}
    assert d.conv_offset_mask.weight.shape == (27, 8, 3, 3)
    assert float(d.conv_offset_mask.weight.abs().sum()) == 0.0


def test_opts_polydet_heads_and_chunks():
    from centerpoly_amd.opts import opts
    o = opts().init(["polydet", "--arch", "dla_34", "--nbr_points", "32", "--gpus", "0,1,2",
                     "--batch_size", "32", "--master_batch_size", "8"])
    assert o.heads == {"hm": 8, "poly": 64, "pseudo_depth": 1, "reg": 2}
    assert o.head_conv == 256 and o.pad == 31 and o.num_stacks == 1
    assert o.chunk_sizes == [8, 12, 12]
    o = opts().init(["polydet", "--arch", "hourglass"])
    assert o.head_conv == 64 and o.pad == 127 and o.num_stacks == 2


def test_post_process_mirror_matches_golden(golden):
    from centerpoly_amd.utils.post_process import polydet_post_process
    from oracle import decode as odec
    g = golden("post_cart16")
    case = cases.DECODE_CASES[0]
    name, B, C, h, w, N, K, rep = case
    heat, polys, depth, reg = (torch.from_numpy(a) for a in cases.decode_inputs_np(*case))
    dets, _, _ = odec.polydet_decode(heat, polys, depth, reg, K=K, rep=rep)
    ret = polydet_post_process(dets.numpy()[:1].copy(), [cases.POST_META["c"]],
                               [cases.POST_META["s"]], h, w, C)
    for j in range(1, C + 1):
        a = np.array(ret[0][j], dtype=np.float32).reshape(-1, 2 * N + 6)
        np.testing.assert_allclose(a, g["cls%d" % j], rtol=1e-6, atol=1e-4)


def test_ctypes_signatures_match_header_arity():
    """Every declaration of include/centerpoly_hip.h is bound in _C._SIGNATURES with the same
    number of arguments (a drifted binding would corrupt the call instead of failing)."""
    import re
    from centerpoly_amd import _C
    text = open(os.path.join(ROOT, "include", "centerpoly_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = re.findall(r"\b(?:int|size_t|const char\*)\s+(cp_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S)
    assert len(decls) >= 30
    for name, args in decls:
        n = 0 if args.strip() in ("", "void") else len(args.split(","))
        assert name in _C._SIGNATURES, name
        assert len(_C._SIGNATURES[name][1]) == n, name
