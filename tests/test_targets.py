"""Training-target construction: oracle/targets.py against the reference's own helpers
(tests/golden/targets_prims.npz, CPU) and the HIP kernels behind cp_polydet_targets against
the oracle (GPU: indices / masks bit-exact, float targets to 1 ulp of fp32)."""
import numpy as np
import pytest
import torch

from centerpoly_amd import synth
from oracle import post as opost
from oracle import targets as otg


def test_helpers_match_reference_golden(golden):
    g = golden("targets_prims")
    for (hh, ww), r in zip(g["radius_sizes"], g["radius"]):
        assert otg.gaussian_radius((int(hh), int(ww))) == r
    hm = np.zeros((48, 64), dtype=np.float32)
    centers = synth.integers("targets/centers", (12, 2), 0, 48)
    radii = synth.integers("targets/radii", (12,), 0, 9)
    centers[0] = (0, 0)
    centers[1] = (63, 47)
    for (cx, cy), r in zip(centers, radii):
        otg.draw_umich_gaussian(hm, (int(min(cx + 8, 63)), int(cy)), int(r))
    assert np.array_equal(hm, g["splat_hm"])
    pts = synth.uniform("targets/pts", (32, 2), -50.0, 2100.0)
    got = np.stack([otg.affine_transform(p, g["affine_t"]) for p in pts])
    assert np.array_equal(got, g["affine"])


def _case(tag, in_h, in_w, N, rep, flipped, scale=1.0, n_objs=None):
    anns = synth.raw_annotations("tg/" + tag, in_h, in_w, nbr_points=N, n_objs=n_objs)
    c = np.array([in_w * 0.47, in_h * 0.55], np.float32)
    s = max(in_h, in_w) * scale
    oh, ow = in_h // 4, in_w // 4
    t = opost.get_affine_transform(c, s, 0, [ow, oh])
    return anns, t, flipped, in_w, oh, ow, N, rep


def test_oracle_targets_invariants():
    anns, t, flipped, width, oh, ow, N, rep = _case("inv", 512, 1024, 16, "cartesian", False)
    r = otg.build_targets(anns, t, flipped, width, oh, ow, 8, 128, N, rep)
    n = int(r["reg_mask"].sum())
    assert 0 < n <= len(anns)
    k = np.nonzero(r["reg_mask"])[0]
    cy, cx = r["ind"][k] // ow, r["ind"][k] % ow
    for kk, y, x in zip(k, cy, cx):
        assert r["hm"][anns[kk]["cls_id"], y, x] == 1.0        # exact 1 at every centre
        np.testing.assert_allclose(r["peak"][kk], np.array([x, y]) + r["reg"][kk], rtol=0, atol=1e-5)
    assert (r["reg"] >= 0).all() and (r["reg"] < 1).all()
    assert r["hm"].max() == 1.0 and r["border_hm"].max() == 1.0
    # mirrored twice with the vertex re-ordering the polygon keeps its cyclic order
    rf = otg.build_targets(anns, t, True, width, oh, ow, 8, 128, N, rep)
    assert int(rf["reg_mask"].sum()) > 0 and not np.array_equal(rf["poly"], r["poly"])
    # an image without objects: neutral frequency weight
    assert otg.build_targets([], t, False, width, oh, ow, 8, 128, N, rep)["freq_mask"] == 1.0


GPU_CASES = [
    ("a", 512, 1024, 16, "cartesian", False, 1.0, None),
    ("b", 512, 1024, 16, "cartesian", True, 0.7, None),
    ("c", 384, 1280, 32, "cartesian", True, 1.3, 40),
    ("d", 512, 1024, 24, "polar", False, 0.9, None),
    ("e", 512, 1024, 24, "polar", True, 1.0, 25),
    ("f", 256, 256, 16, "polar_fixed", True, 0.6, 12),
]


@pytest.mark.gpu
def test_targets_kernel_vs_oracle_batch():
    from centerpoly_amd.datasets.sample.polydet import build_targets, collate, pack_annotations
    for rep in ("cartesian", "polar"):
        cases = [c for c in GPU_CASES if c[4] == rep and c[1:3] == (512, 1024)]
        N = cases[0][3]
        packed, refs = [], []
        for tag, in_h, in_w, n, rp, flipped, scale, n_objs in cases:
            if n != N:
                continue
            anns, t, fl, width, oh, ow, _, _ = _case(tag, in_h, in_w, n, rp, flipped, scale, n_objs)
            packed.append(pack_annotations(anns, t, fl, width, 128, N))
            refs.append(otg.build_targets(anns, t, fl, width, oh, ow, 8, 128, N, rp))
        raw = {k: v.cuda() for k, v in collate(packed).items()}
        out = {k: v.cpu().numpy() for k, v in build_targets(raw, 128, 256, 8, rep=rep).items()}
        for b, ref in enumerate(refs):
            _compare(out, b, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_targets_kernel_vs_oracle_single(case):
    from centerpoly_amd.datasets.sample.polydet import build_targets, collate, pack_annotations
    tag, in_h, in_w, N, rep, flipped, scale, n_objs = case
    anns, t, fl, width, oh, ow, _, _ = _case(tag, in_h, in_w, N, rep, flipped, scale, n_objs)
    ref = otg.build_targets(anns, t, fl, width, oh, ow, 8, 128, N, rep)
    raw = {k: v.cuda() for k, v in collate([pack_annotations(anns, t, fl, width, 128, N)]).items()}
    out = {k: v.cpu().numpy() for k, v in build_targets(raw, oh, ow, 8, rep=rep).items()}
    _compare(out, 0, ref)


def _ulp_close(a, b, ulps=1):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    tol = ulps * np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32))
    return np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= tol)


def _compare(out, b, ref):
    assert np.array_equal(out["reg_mask"][b], ref["reg_mask"])
    assert np.array_equal(out["ind"][b], ref["ind"])
    assert np.array_equal(out["pseudo_depth"][b], ref["pseudo_depth"])
    for k in ("peak", "reg", "wh", "poly"):
        assert _ulp_close(out[k][b], ref[k]), k
    # heat maps: same support, values to 1 ulp (device exp vs numpy exp before the fp32 cast)
    for k in ("hm", "border_hm"):
        assert np.array_equal(out[k][b] > 0, ref[k] > 0), k
        assert _ulp_close(out[k][b], ref[k]), k
        assert np.array_equal(out[k][b] == 1.0, ref[k] == 1.0), k
    np.testing.assert_allclose(out["freq_mask"][b], ref["freq_mask"], rtol=1e-6)


@pytest.mark.gpu
def test_targets_feed_the_loss():
    """Targets built on the device drive PolydetLoss end to end (finite loss, non-zero grads)."""
    from centerpoly_amd.datasets.sample.polydet import build_targets, collate, pack_annotations
    from centerpoly_amd.opts import opts
    from centerpoly_amd.trains.polydet import PolydetLoss
    opt = opts().init(["polydet", "--arch", "smallhourglass", "--poly_loss", "l1+iou", "--nbr_points", "16"])
    opt.device = torch.device("cuda")
    anns, t, fl, width, oh, ow, N, rep = _case("loss", 256, 256, 16, "cartesian", False, 1.0, 10)
    raw = {k: v.cuda() for k, v in collate([pack_annotations(anns, t, fl, width, 128, N)]).items()}
    batch = build_targets(raw, oh, ow, 8, rep=rep)
    g = torch.Generator().manual_seed(1)
    leaves = {"hm": torch.randn(1, 8, oh, ow, generator=g).cuda().requires_grad_(),
              "poly": torch.randn(1, 32, oh, ow, generator=g).cuda().requires_grad_(),
              "pseudo_depth": torch.randn(1, 1, oh, ow, generator=g).cuda().requires_grad_(),
              "reg": torch.randn(1, 2, oh, ow, generator=g).cuda().requires_grad_()}
    outputs = [{k: v * 1.0 for k, v in leaves.items()}]      # the loss applies its sigmoid in place
    loss, stats = PolydetLoss(opt)(outputs, batch)
    assert torch.isfinite(loss)
    loss.backward()
    assert float(leaves["hm"].grad.abs().sum()) > 0 and float(leaves["poly"].grad.abs().sum()) > 0


@pytest.mark.gpu
def test_trainer_epoch_with_device_targets():
    """--device_targets: loader packs raw annotations, PolydetTrainer.prepare_batch builds the
    targets on the GPU, two optimisation steps run."""
    import contextlib
    import io
    from centerpoly_amd.datasets.dataset_factory import get_dataset
    from centerpoly_amd.models.model import create_model
    from centerpoly_amd.opts import opts
    from centerpoly_amd.trains.train_factory import train_factory
    with contextlib.redirect_stdout(io.StringIO()):
        opt = opts().init(["polydet", "--arch", "smallhourglass", "--device_targets", "--input_h", "256",
                           "--input_w", "256", "--batch_size", "2", "--num_iters", "2",
                           "--poly_loss", "l1+iou"])
        Dataset = get_dataset("synthetic", opt.task)          # (the real dataset names need files)
        opt = opts().update_dataset_info_and_set_heads(opt, Dataset)
        ds = Dataset(opt, "train")
    opt.device = torch.device("cuda")
    torch.manual_seed(317)
    model = create_model(opt.arch, opt.heads, opt.head_conv)
    trainer = train_factory["polydet"](opt, model, torch.optim.Adam(model.parameters(), opt.lr))
    trainer.set_device(opt.gpus, opt.chunk_sizes, opt.device)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)
    stats, _ = trainer.train(1, loader)
    assert np.isfinite(stats["loss"]) and stats["hm_l"] > 0 and stats["poly_l"] > 0


@pytest.mark.gpu
def test_targets_edge_cases():
    """No objects, more annotations than slots, and objects that the crop pushes fully outside
    (skipped slots stay zero but keep their index, as in the reference's loop)."""
    from centerpoly_amd.datasets.sample.polydet import build_targets, collate, pack_annotations
    N, M = 16, 8
    t = opost.get_affine_transform(np.array([256., 128.], np.float32), 512.0, 0, [128, 64])
    many = synth.raw_annotations("tg/many", 256, 512, nbr_points=N, n_objs=20)
    far = [dict(a) for a in synth.raw_annotations("tg/far", 256, 512, nbr_points=N, n_objs=5)]
    for a in far[1:4]:                                       # move three objects far outside the image
        a["bbox"] = [a["bbox"][0] + 5000.0, a["bbox"][1], a["bbox"][2], a["bbox"][3]]
        a["poly"] = [v + 5000.0 if i % 2 == 0 else v for i, v in enumerate(a["poly"])]
    cases = [[], many, far]
    raw = {k: v.cuda() for k, v in collate([pack_annotations(c, t, False, 512, M, N) for c in cases]).items()}
    out = {k: v.cpu().numpy() for k, v in build_targets(raw, 64, 128, 8).items()}
    for b, anns in enumerate(cases):
        _compare(out, b, otg.build_targets(anns, t, False, 512, 64, 128, 8, M, N, "cartesian"))
    assert out["reg_mask"][0].sum() == 0 and out["freq_mask"][0] == 1.0 and out["hm"][0].max() == 0.0
    assert out["reg_mask"][1].sum() <= M
    assert list(out["reg_mask"][2][1:4]) == [0, 0, 0]


# ---------------------------------------------------------------------------------------------------------------------
# The object loop against the REFERENCE's own PolydetDataset.__getitem__ (tests/golden/sampler_*.npz, made by
# tests/golden/gen_sampler_golden.py running src/lib/datasets/sample/polydet.py:66-449 on KITTIPolyStuff/BBoxes/val16.json)
SAMPLER_CASES = ["cart_crop", "cart_flip", "cart_shift", "cart_noreorder", "polar_flip", "polar_fixed", "cart_val",
                 "cart_dense", "cart_catspec", "polar_catspec"]       # the last three: --dense_poly / --cat_spec_poly
CLASS_NAMES = ["person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle"]


def _border(border, size):
    i = 1
    while size - border // i <= border // i:
        i *= 2
    return border // i


def _replay_draws(g, n):
    """The sampler's random draws in the reference's order (sample/polydet.py:94-113): scale choice, centre x, centre y
    (or the two shifts and the scale of --not_rand_crop), then the flip -- from the fixture's seed."""
    H, W = [int(v) for v in g["img_hw"]]
    np.random.seed(int(g["seed"]))
    out = []
    for _ in range(n):
        c = np.array([W / 2., H / 2.], dtype=np.float32)
        s = max(H, W) * 1.0
        flipped = False
        if str(g["split"]) == "train":
            if not bool(g["not_rand_crop"]):
                s = s * np.random.choice(np.arange(0.6, 1.4, 0.1))
                wb, hb = _border(128, W), _border(128, H)
                c[0] = np.random.randint(low=wb, high=W - wb)
                c[1] = np.random.randint(low=hb, high=H - hb)
            else:
                sf, cf = float(g["scale"]), float(g["shift"])
                c[0] += s * np.clip(np.random.randn() * cf, -2 * cf, 2 * cf)
                c[1] += s * np.clip(np.random.randn() * cf, -2 * cf, 2 * cf)
                s = s * np.clip(np.random.randn() * sf + 1, 1 - sf, 1 + sf)
            if np.random.random() < float(g["flip_prob"]):
                flipped = True
                c[0] = W - c[0] - 1
        out.append((c, s, flipped))
    return out


def _sampler_anns(g, j):
    freq = g["class_freq"]
    return [{"bbox": list(b), "poly": list(p), "cls_id": int(c) - 1, "pseudo_depth": float(d), "freq": float(freq[int(c) - 1])}
            for b, p, c, d in zip(g["s%d_ann_bbox" % j], g["s%d_ann_poly" % j], g["s%d_ann_cat" % j], g["s%d_ann_depth" % j])]


@pytest.mark.parametrize("case", SAMPLER_CASES)
def test_object_loop_matches_reference_sampler(case, golden):
    """oracle/targets.py::build_targets == the reference's own __getitem__, array for array, and the replayed draws
    (what centerpoly_amd's host sampler mirrors) give the reference's centre and scale."""
    g = golden("sampler_" + case)
    n = len(g["img_ids"])
    oh, ow = [int(v) for v in g["out_hw"]]
    W = int(g["img_hw"][1])
    for j, (c, s, flipped) in enumerate(_replay_draws(g, n)):
        assert np.array_equal(c, g["s%d_c" % j]) and float(s) == float(g["s%d_s" % j]), (j, c, s)
        t = opost.get_affine_transform(c, s, 0, [ow, oh])
        dense, catspec = bool(g.get("dense_poly", False)), bool(g.get("cat_spec_poly", False))
        r = otg.build_targets(_sampler_anns(g, j), t, flipped, W, oh, ow, 8, 128, 16, str(g["rep"]),
                              no_reorder_flip=bool(g["no_reorder_flip"]), dense_poly=dense, cat_spec_poly=catspec)
        if "s%d_keys" % j in g:                       # the dict's key set (cat-spec: no freq_mask / border_hm / wh; dense: no poly)
            assert sorted(r) == sorted(str(k) for k in g["s%d_keys" % j]), (case, sorted(r))
        for k in ("hm", "reg_mask", "ind", "poly", "pseudo_depth", "border_hm", "wh", "peak", "reg", "cat_spec_poly",
                  "cat_spec_mask", "dense_poly", "dense_poly_mask"):
            assert (k in r) == ("s%d_%s" % (j, k) in g), (case, j, k)
            if k in r:
                assert np.array_equal(r[k], g["s%d_%s" % (j, k)]), (case, j, k)
                assert r[k].dtype == g["s%d_%s" % (j, k)].dtype, (case, j, k)
        if "freq_mask" in r:
            assert float(r["freq_mask"]) == float(g["s%d_freq_mask" % j])
        assert int(r["reg_mask"].sum()) > 0
        if dense:
            assert 0 < float(r["dense_poly_mask"].mean()) < 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("case", SAMPLER_CASES)
def test_device_targets_match_reference_sampler(case, golden):
    """cp_polydet_targets (csrc/targets.hip) against the arrays the reference's sampler produced: indices, masks and
    heat maps bit-exact, float targets to 1 ulp of fp32 (the kernel's fp32 stores of float64 arithmetic)."""
    from centerpoly_amd.datasets.sample.polydet import build_targets, collate, pack_annotations
    g = golden("sampler_" + case)
    n = len(g["img_ids"])
    oh, ow = [int(v) for v in g["out_hw"]]
    W = int(g["img_hw"][1])
    packed = []
    for j, (c, s, flipped) in enumerate(_replay_draws(g, n)):
        t = opost.get_affine_transform(c, s, 0, [ow, oh])
        packed.append(pack_annotations(_sampler_anns(g, j), t, flipped, W, 128, 16))
    raw = {k: v.cuda() for k, v in collate(packed).items()}
    dense, catspec = bool(g.get("dense_poly", False)), bool(g.get("cat_spec_poly", False))
    out = build_targets(raw, oh, ow, 8, rep=str(g["rep"]), no_reorder_flip=bool(g["no_reorder_flip"]), dense_poly=dense,
                        cat_spec_poly=catspec)
    for j in range(n):
        if "s%d_keys" % j in g:
            assert sorted(out) == sorted(str(k) for k in g["s%d_keys" % j]), (case, sorted(out))
        for k in ("reg_mask", "ind", "hm", "border_hm", "cat_spec_mask"):
            if k in out:
                assert np.array_equal(out[k][j].cpu().numpy(), g["s%d_%s" % (j, k)]), (case, j, k)
        for k in ("poly", "pseudo_depth", "wh", "peak", "reg", "cat_spec_poly"):
            if k in out:
                ref = g["s%d_%s" % (j, k)]
                np.testing.assert_allclose(out[k][j].cpu().numpy(), ref, rtol=2.4e-7, atol=1e-6 * max(1.0, np.abs(ref).max()),
                                           err_msg="%s %d %s" % (case, j, k))
        if "freq_mask" in out:
            np.testing.assert_allclose(float(out["freq_mask"][j]), float(g["s%d_freq_mask" % j]), rtol=1e-6)
        if dense:
            # which object owns a pixel is decided by `float64 Gaussian >= float32 map` (utils/image.py:201): the kernel's
            # exp may differ from numpy's in the last bit of the double, which flips that test only where the Gaussian sits
            # within one ulp of a float32 grid point -- allow a handful of such pixels, everything else is exact
            ref_m, got_m = g["s%d_dense_poly_mask" % j], out["dense_poly_mask"][j].cpu().numpy()
            ref_d, got_d = g["s%d_dense_poly" % j], out["dense_poly"][j].cpu().numpy()
            px_diff = (ref_m != got_m).any(axis=0)
            assert px_diff.sum() <= 2, (case, j, int(px_diff.sum()))
            same = ~px_diff
            np.testing.assert_allclose(got_d[:, same], ref_d[:, same], rtol=2.4e-7, atol=1e-5)
            assert ref_m.sum() > 0 and set(np.unique(got_m)) <= {0.0, 1.0}
