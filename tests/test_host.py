"""CPU: host-side logic and the C-ABI surface (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from fgvc_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "fgvc_hip.h")).read()
    declared = set(re.findall(r"\b(fgvc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fgvc_version().startswith(b"fgvc_hip")


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (error codes, no exceptions across the ABI)."""
    from fgvc_amd import _lib
    lib = _lib.load()
    rc = lib.fgvc_pair_topk_f32(None, None, None, 1, 256, 4, 4, 4, 4, 1, 1, 1, 10, None, None, None, None)
    assert rc == 1 and b"null pointer" in lib.fgvc_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.fgvc_pair_topk_f32(p, p, p, 1, 256, 4, 4, 4, 4, 1, 1, 1, 17, None, p, p, None)
    assert rc == 2 and b"topk" in lib.fgvc_last_error()
    rc = lib.fgvc_pair_topk_f32(p, p, p, 1, 256, 4, 4, 5, 4, 224, _lib.NO_LIMIT, _lib.NO_LIMIT, 10, None, p, p, None)
    assert rc == 1 and b"equal query/key grids" in lib.fgvc_last_error()
    rc = lib.fgvc_merge_topk_f32(p, p, p, 1, 1, 4, 4, 10, ctypes.c_float(0.0), 0, p, p, p, None)
    assert rc == 1 and b"temperature" in lib.fgvc_last_error()
    rc = lib.fgvc_corr_volume_bf16x3(p, p, 48, 4, 4, ctypes.c_float(1.0), p, None)
    assert rc == 2 and b"multiple of 64" in lib.fgvc_last_error()


def test_r2max_matches_oracle():
    from fgvc_amd import _lib
    from oracle import fgvc_oracle as O
    lib = _lib.load()
    for r in [0.5, 1, 1.5, 2, 2.5, 3, 7, 12, 15, 15.5, 24, 100]:
        assert lib.fgvc_r2max_for_radius(r) == O.radius_predicate_r2max(r), r


def test_ops_refuse_cpu_tensors():
    from fgvc_amd import _lib, ops
    with pytest.raises(_lib.FgvcHipError):
        ops.normalize_to_hwc(torch.zeros(1, 32, 4, 4))


def test_key_slots_and_plan():
    from fgvc_amd import engine
    from oracle import fgvc_oracle as O
    cfg = engine.TrackerConfig()
    for f in range(1, 12):
        assert engine.key_slots(f, 0, 5, True) == O.key_slots(f, 5, True)
        assert engine.key_slots(f + 3, 3, 5, True) == [k + 3 for k in O.key_slots(f, 5, True)]
    plan = engine.plan_clip(8, [0], cfg)
    assert len(plan.pairs) == 27 and len(plan.slot_pair) == 7 and plan.t_max == 6   # 32 slots, 5 duplicates of frame 0
    assert sum(1 for row in plan.slot_pair for p in row if p >= 0) == 32
    # frame 3: slots [0,0,1,2] -> the two frame-0 slots share one pair
    row = plan.out_rows[(0, 3)]
    assert plan.slot_frame[row][:4] == [0, 0, 1, 2] and plan.slot_pair[row][0] == plan.slot_pair[row][1]
    assert all(m for (_, _, m) in plan.pairs)
    # several query times: pairs are shared between groups
    p2 = engine.plan_clip(12, [0, 4], cfg)
    single = len(engine.plan_clip(12, [0], cfg).pairs) + len(engine.plan_clip(8, [0], cfg).pairs)
    assert len(p2.pairs) < single
    # with_first_neighbor=False: slot 0 is unmasked, and (f,0) may exist in both flavours
    p3 = engine.plan_clip(4, [0], engine.TrackerConfig(with_first_neighbor=False))
    assert (1, 0, False) in p3.pairs and (1, 0, True) in p3.pairs
    # no mask at all
    p4 = engine.plan_clip(4, [0], engine.TrackerConfig(neighbor_range=None))
    assert not any(m for (_, _, m) in p4.pairs)
    # frame_range restricts the query frames
    p5 = engine.plan_clip(20, [0], cfg, frame_range=(10, 15))
    assert sorted({q for (q, _, _) in p5.pairs}) == [10, 11, 12, 13, 14]


def test_registry_builder_config(tmp_path):
    import fgvc_amd.mmpt_api as api
    assert {"VanillaTracker", "HRVanillaTracker", "BaseTracker"} <= set(api.MODELS.module_dict)
    assert "ResNet" in api.BACKBONES
    with pytest.raises(KeyError):
        api.build_backbone(dict(type="NoSuchNet"))
    base = tmp_path / "base.py"
    base.write_text("data = dict(a=1, b=dict(c=2, d=3))\nx = 5\n")
    child = tmp_path / "child.py"
    child.write_text("_base_ = './base.py'\ndata = dict(b=dict(c=7))\nexp_name = 'e'\nwork_dir = f'./eval/{exp_name}'\n")
    cfg = api.Config.fromfile(str(child))
    assert cfg.data.b.c == 7 and cfg.data.b.d == 3 and cfg.x == 5 and cfg.work_dir == "./eval/e"
    assert cfg.get("eval_arc", "VanillaTracker") == "VanillaTracker"


def test_reference_eval_config_loads():
    """the shipped eval config of the reference parses with our Config (only if the tree is present)."""
    path = "/root/reference/configs/eval/res18_d1_eval.py"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    import fgvc_amd.mmpt_api as api
    cfg = api.Config.fromfile(path)
    assert cfg.model.type == "VanillaTracker" and cfg.test_cfg_davis.neighbor_range == 30
    tc = cfg["test_cfg_davis"]
    model = api.build_model(dict(type=cfg.get("eval_arc", "VanillaTracker"),
                                 backbone=dict(cfg.model.backbone, out_indices=tc.out_indices, strides=tc.strides)),
                            train_cfg=None, test_cfg=tc)
    ec = model.engine_config()
    assert (ec.precede_frames, ec.topk, ec.temperature, ec.neighbor_range) == (5, 10, 0.07, 30)


def test_resnet_state_dict_names_and_arithmetic():
    """key names equal the reference's (the oracle ResNet was loaded strict=True into the reference backbone
    by the golden generator) and the forward pass equals the oracle's."""
    import fgvc_amd.mmpt_api as api
    from oracle import fgvc_oracle as O
    net = api.build_backbone(dict(type="ResNet", depth=18, strides=(1, 1, 1, 4), out_indices=(2,), pool_type="none"))
    ora = O.ResNet18((1, 1, 1, 4), 2, "none")
    assert set(net.state_dict()) == set(ora.state_dict())
    sd = O.seeded_resnet_state(3, (1, 1, 1, 4), "none")
    net.load_state_dict(sd)
    ora.load_state_dict(sd)
    x = torch.randn(2, 3, 48, 64)
    with torch.no_grad():
        a, b = net.eval()(x), ora.eval()(x)
    assert a.shape == (2, 256, 24, 32) and torch.allclose(a, b, atol=1e-5, rtol=1e-5)
    # checkpoint with the tracker's `backbone.` prefix
    api.load_checkpoint(net, {"state_dict": {"backbone." + k: v for k, v in sd.items()}})


def test_neighbor_mask_dense_matches_reference_golden(golden):
    import numpy as np
    import fgvc_amd.mmpt_api as api
    for name, mode in [("mask_circle_8x12_r6", "circle"), ("mask_square_9x7_r5", "square"),
                       ("mask_circle_20x24_r14", "circle")]:
        g = golden(name)
        H, W, nr = int(g["H"]), int(g["W"]), int(g["nr"])
        m = api.common.spatial_neighbor(1, H, W, nr, "cpu", torch.float32, mode=mode)
        ref = np.unpackbits(g["packed"])[: (H * W) ** 2].reshape(H * W, H * W).astype(bool)
        assert m.shape == (H * W, H * W)
        assert np.array_equal(m.dense().numpy(), ref)


def test_install_as_mmpt():
    import subprocess
    import sys
    code = ("import fgvc_amd; fgvc_amd.install_as_mmpt();"
            "from mmpt.models import build_model, MODELS;"
            "from mmpt.models.common import masked_attention_efficient, spatial_neighbor;"
            "from mmpt.models.registry import BACKBONES; from mmpt.apis import single_gpu_test, multi_gpu_test;"
            "print(sorted(MODELS.module_dict))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr
    assert "VanillaTracker" in out.stdout


def test_shard_frames():
    from fgvc_amd import dist as D
    from fgvc_amd.engine import TrackerConfig
    assert D.shard_frames(9, 2) == [(1, 5), (5, 9)]
    assert D.shard_frames(3, 4) == [(1, 2), (2, 3), (3, 3), (3, 3)]
    r = D.shard_frames(64, 8)
    assert r[0][0] == 1 and r[-1][1] == 64 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    cfg = TrackerConfig()
    assert D.encode_range(9, 17, [0], cfg) == (4, 17)
    assert D.encode_range(1, 9, [0], cfg) == (0, 9)
    assert D.encode_range(5, 5, [0], cfg) == (5, 5)
    enc = [D.encode_range(lo, hi, [0, 20], cfg) for lo, hi in D.shard_frames(64, 8)]
    assert D.owner_of(0, enc) == 0 and D.owner_of(20, enc) == 2


def test_prepare_conv_split_layout_reconstructs_conv_bn():
    """Host side of fgvc_conv_split_f32 (no GPU): BatchNorm folding and the packed weight layout
    [tap = ky*KS+kx][Cin/32][Cout][hi 32 ci | lo 32 ci].  Unpacking hi + lo and running a plain convolution must
    reproduce conv -> BN(eval); the padded activation dims follow the documented formula."""
    import torch.nn.functional as F
    from fgvc_amd import ops
    g = torch.Generator().manual_seed(3)
    Cin, Cout, KS = 64, 128, 3
    w = torch.randn(Cout, Cin, KS, KS, generator=g) * 0.1
    bn = torch.nn.BatchNorm2d(Cout).eval()
    bn.weight.data = torch.rand(Cout, generator=g) + 0.5
    bn.bias.data = torch.randn(Cout, generator=g)
    bn.running_mean = torch.randn(Cout, generator=g)
    bn.running_var = torch.rand(Cout, generator=g) + 0.5
    packed, bias = ops.prepare_conv_split(w, bn)
    assert packed.shape == (KS * KS, Cin // 32, Cout, 64) and packed.dtype == torch.int16 and bias.shape == (Cout,)
    v = packed.view(torch.bfloat16).float()
    wrec = (v[..., :32] + v[..., 32:])                                  # [tap][chunk][co][32 ci]
    wrec = wrec.permute(2, 1, 3, 0).reshape(Cout, Cin, KS, KS)           # tap -> (ky, kx), (chunk, ci) -> input channel
    x = torch.randn(2, Cin, 9, 11, generator=g)
    want = bn(F.conv2d(x, w, padding=1)).detach()
    got = F.conv2d(x, wrec, bias, padding=1)
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())
    # lo really is the residue of hi (|lo| <= 2^-8 |hi|), i.e. the split is not two independent roundings of w
    assert float((v[..., 32:].abs() - v[..., :32].abs() * 2.0 ** -7).clamp_min(0).max()) == 0.0
    assert ops.conv_pad_dims(120, 214) == (122, 232) and ops.conv_pad_dims(8, 32) == (10, 40) and ops.conv_pad_dims(9, 33) == (18, 72)
    assert ops.split_path_ok(256, 120, 214, 10, True) and not ops.split_path_ok(256, 120, 214, 11, True)


def test_rgb_to_lab_known_answers_and_input_contract():
    """F3 input contract: CIE L*a*b* of the sRGB primaries (published values), grey axis, and the normalisation of
    configs/eval/base_data.py:1-7."""
    from fgvc_amd.datasets import rgb_to_lab, preprocess_tapvid_frames
    cols = torch.tensor([[1.0, 1, 1], [0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.5, 0.5, 0.5]]).t().reshape(3, 1, 6)
    lab = rgb_to_lab(cols)[:, 0].t()
    want = torch.tensor([[100.0, 0, 0], [0, 0, 0], [53.2408, 80.0925, 67.2032], [87.7347, -86.1827, 83.1793],
                         [32.2970, 79.1875, -107.8602], [53.3890, 0, 0]])
    assert torch.allclose(lab, want, atol=0.02), (lab - want).abs().max()
    frames = torch.randint(0, 256, (3, 40, 60, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    x = preprocess_tapvid_frames(frames, size=(32, 48))
    assert x.shape == (1, 3, 3, 32, 48) and x.dtype == torch.float32
    assert float(x[:, :, 0].min()) >= -1.0 - 1e-6 and float(x[:, :, 0].max()) <= 1.0 + 1e-6      # L in [0,100] -> [-1,1]
    assert float(x[:, :, 1:].abs().max()) <= 1.01                                               # a,b within +-127
    grey = torch.full((1, 8, 8, 3), 128, dtype=torch.uint8)
    xg = preprocess_tapvid_frames(grey, size=(8, 8))
    assert float(xg[0, 0, 1:].abs().max()) < 1e-4                                              # neutral grey: a = b = 0


def test_resnet_split_cache_is_dropped_when_weights_change():
    """The folded / split convolution weights cached by the backbone must not survive a (parent) load_state_dict."""
    import fgvc_amd.mmpt_api as api
    model = api.build_model(dict(type="VanillaTracker",
                                 backbone=dict(type="ResNet", depth=18, strides=(1, 2, 1, 1), out_indices=(2,),
                                               pool_type="none")),
                            train_cfg=None, test_cfg=api.ConfigDict(topk=10))
    bb = model.backbone
    bb.__dict__["_split_cache"] = {"stale": 1}
    model.load_state_dict(model.state_dict())              # parent-level load reaches the backbone's hook
    assert "_split_cache" not in bb.__dict__
    bb.__dict__["_split_cache"] = {"stale": 1}
    bb.init_weights()
    assert "_split_cache" not in bb.__dict__


def test_tapvid_pickles_sample_format(tmp_path):
    """TapVidPickles on files in the TAP-Vid layout (per-video pickles, one multi-video pickle, JPEG-encoded frames): the sample
    format and conventions of the reference's TAPVidDataset (tapvid.py:85-174) -- (t,x,y) queries in input_size pixels, 'first'
    drops never-visible tracks and queries at the first visible frame, 'strided' repeats tracks per query frame."""
    import io, pickle
    import numpy as np
    from PIL import Image
    from fgvc_amd.datasets import TapVidPickles
    rng = np.random.default_rng(0)
    T, H, W, P = 7, 40, 60, 5
    video = rng.integers(0, 255, (T, H, W, 3), dtype=np.uint8)
    points = rng.random((P, T, 2)).astype(np.float32)
    occ = np.zeros((P, T), dtype=bool)
    occ[1, :3] = True            # first visible at t = 3
    occ[2, :] = True             # never visible: dropped by 'first'
    occ[3, 5] = True             # hidden at a strided query frame
    sample = dict(video=video, points=points, occluded=occ)
    d = tmp_path / "videos"
    d.mkdir()
    for name in ("a", "b"):
        with open(d / f"{name}.pkl", "wb") as f:
            pickle.dump(sample, f)
    ds = TapVidPickles(str(d), "first", (32, 48))
    assert len(ds) == 2
    s0 = ds[0]
    assert s0["rgbs"].shape == (1, T, 3, 32, 48) and s0["rgbs"].dtype == torch.float32
    assert s0["query_points"].shape == (1, 4, 3) and s0["trajectories"].shape == (1, T, 4, 2) and s0["visibilities"].shape == (1, T, 4)
    qp = s0["query_points"][0]
    assert qp[:, 0].tolist() == [0.0, 3.0, 0.0, 0.0]
    assert torch.allclose(qp[1, 1:], torch.tensor(points[1, 3] * np.array([48, 32], dtype=np.float32)))     # (x, y) in pixels
    assert torch.allclose(s0["trajectories"][0, :, 0], torch.from_numpy(points[0] * np.array([48, 32], dtype=np.float32)))
    assert s0["visibilities"][0, :, 1].tolist() == [0, 0, 0, 1, 1, 1, 1]
    st = TapVidPickles(str(d), "strided", (32, 48))[1]
    # query frames 0 and 5: visible tracks {0, 3, 4} at t = 0 and {0, 1, 4} at t = 5
    assert st["query_points"][0][:, 0].tolist() == [0, 0, 0, 5, 5, 5]
    assert torch.allclose(st["trajectories"][0, :, 3], torch.from_numpy(points[0] * np.array([48, 32], dtype=np.float32)))
    # one pickle with several videos, frames stored as JPEG bytes
    jpeg = []
    for t in range(T):
        b = io.BytesIO()
        Image.fromarray(video[t]).save(b, format="JPEG")
        jpeg.append(b.getvalue())
    big = tmp_path / "tapvid_multi.pkl"
    with open(big, "wb") as f:
        pickle.dump({"v0": sample, "v1": dict(video=np.array(jpeg, dtype=object), points=points, occluded=occ)}, f)
    dm = TapVidPickles(str(big), "first", (32, 48))
    assert len(dm) == 2 and dm[1]["rgbs"].shape == (1, T, 3, 32, 48) and torch.equal(dm[0]["rgbs"], s0["rgbs"])
    with pytest.raises(ValueError):
        TapVidPickles(str(d), "random")


def test_torchvision_checkpoint_key_map_matches_reference():
    """ResNet.init_weights with a torchvision checkpoint (the constructor default, resnet.py:566-585): every own tensor is filled
    from the checkpoint key the genuine loader used (tests/golden/tv_keymap.json, recorded from the reference)."""
    import json
    from fgvc_amd.mmpt_api import backbones
    keymap = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tv_keymap.json")))
    net = backbones.ResNet(depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none")
    assert set(net.state_dict()) == set(keymap)
    tv = {}
    for k, src in keymap.items():
        assert src is None or backbones.torchvision_key(k) == src, (k, src, backbones.torchvision_key(k))
        if src is not None:
            tv[src] = torch.full_like(net.state_dict()[k], float(len(tv) + 1))
    tv["fc.weight"] = torch.zeros(1000, 512)
    net.pretrained = tv                                   # a state dict, torchvision_pretrain=True (:586-590)
    net.init_weights()
    sd = net.state_dict()
    for k, src in keymap.items():
        if src is not None:
            assert torch.equal(sd[k], tv[src].to(sd[k].dtype)), k
    net2 = backbones.ResNet(depth=18, pretrained=dict(tv), torchvision_pretrain=False)
    with pytest.raises(Exception):
        net2.init_weights()                               # a dict is only accepted as a torchvision checkpoint (:590)


def test_test_cfg_keys_are_honoured_or_refused():
    """engine.TrackerConfig.from_test_cfg reads the keys as the reference's driver does (vanilla_tracker.py:246, :330-392)."""
    import fgvc_amd.mmpt_api as api
    from fgvc_amd import engine
    c = engine.TrackerConfig.from_test_cfg(api.ConfigDict(precede_frames=3, topk=7, temperature=0.05, neighbor_range=20))
    assert (c.regroup, c.with_first) == (False, True)                      # no key: one group from frame 0, first frame in slot 0
    c = engine.TrackerConfig.from_test_cfg(api.ConfigDict(with_first=True, neighbor_range=20))
    assert (c.regroup, c.with_first) == (True, True)
    c = engine.TrackerConfig.from_test_cfg(api.ConfigDict(with_first=False, neighbor_range=20))
    assert (c.regroup, c.with_first) == (False, False)
    c = engine.TrackerConfig.from_test_cfg(api.ConfigDict(test_mode="v2", neighbor_range=20, mask_mode="square", with_first_neighbor=False,
                                                          sim_mode="l2-distance"))
    assert (c.mask_mode, c.with_first_neighbor, c.sim_mode) == ("circle", True, "dot_product")   # _v2 reads none of the three
    c = engine.TrackerConfig.from_test_cfg(api.ConfigDict(sim_mode="l2-distance", neighbor_range=20))
    assert c.softmax_temperature(256) == 8.0 and engine.TrackerConfig().softmax_temperature(256) == 0.07
    for bad, exc in ((dict(sim_mode="cosine-distance"), NotImplementedError), (dict(sim_mode="l2-distance", with_norm=False), NotImplementedError),
                     (dict(test_mode="v2"), ValueError)):
        with pytest.raises(exc):
            engine.TrackerConfig.from_test_cfg(api.ConfigDict(**bad))


def test_workspace_cache_is_bounded():
    """ResNet keeps the padded activation workspaces of at most `max_workspace_shapes` input shapes (LRU)."""
    from fgvc_amd.mmpt_api import backbones
    net = backbones.ResNet(depth=18, strides=(1, 2, 1, 1), out_indices=(2,), pool_type="none")
    cache = net.__dict__.setdefault("_split_cache", {})
    sigs = [(8, 480, 854, "cpu"), (8, 256, 256, "cpu"), (5, 480, 854, "cpu"), (8, 256, 256, "cpu"), (2, 64, 64, "cpu")]
    for i, sig in enumerate(sigs):
        net._touch_workspace_shape(sig)
        cache[("b", sig, "stage", i)] = object()          # what _split_buffers would add for this shape
        live = {k[1] for k in cache if isinstance(k, tuple) and k[0] == "b"}
        assert len(live) <= net.max_workspace_shapes and sig in live
    assert {k[1] for k in cache if isinstance(k, tuple) and k[0] == "b"} == {(8, 256, 256, "cpu"), (2, 64, 64, "cpu")}


def test_bench_encoder_flops_matches_hand_count():
    """bench.py's FLOP count of the trunk (the numerator of kernels.encoder_phase): ResNet-18 up to stage 3 with strides (1, 2, 1)
    on a 480 x 854 frame, counted by hand layer by layer (resnet.py:457-466, 254-325)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    h1, w1 = 240, 427                      # after the stride-2 stem
    h2, w2 = 120, 214                      # after stage 2's stride
    want = 2.0 * h1 * w1 * 3 * 64 * 49                                        # stem 7x7
    want += 4 * 2.0 * h1 * w1 * 64 * 64 * 9                                   # stage 1: four 3x3 64 -> 64
    want += 2.0 * h2 * w2 * 64 * 128 * 9 + 3 * 2.0 * h2 * w2 * 128 * 128 * 9 + 2.0 * h2 * w2 * 64 * 128      # stage 2 (+ 1x1 projection)
    want += 2.0 * h2 * w2 * 128 * 256 * 9 + 3 * 2.0 * h2 * w2 * 256 * 256 * 9 + 2.0 * h2 * w2 * 128 * 256    # stage 3 (+ 1x1 projection)
    got = bench.encoder_flops(bench.WORKLOADS["cfg2_480p_8f"], 8)
    assert abs(got - 8 * want) < 1e-6 * got


def test_f16_scale_exponents_of_degenerate_tensors():
    """The f16 operand forms scale a tensor by a power of two taken from its largest magnitude.  A zero tensor (zero-initialised
    residual branch, reference resnet.py:596-601) gets exponent 0, tiny / huge tensors are clamped so that activation + weight
    exponents stay inside the C ABI's +-100, inf / NaN raise."""
    import math
    import pytest
    from fgvc_amd import ops
    assert ops.act_scale_log2(0.0) == 0
    assert ops.act_scale_log2(200.0) == 0 and ops.act_scale_log2(256.0) == 0 and ops.act_scale_log2(257.0) == -1
    assert ops.act_scale_log2(1.0) == 8
    assert ops.act_scale_log2(1e-40) == 45 and ops.act_scale_log2(1e30) == -45
    for bad in (math.inf, math.nan):
        with pytest.raises(ValueError):
            ops.act_scale_log2(bad)


def test_f16f6_weight_rows_agree_with_the_oracle_model():
    """The product's host-side packer of the encoder's f16 + FP6 rows (ops._e2m3_blocks / _f16f6_slots: torch, scale from the exponent
    field) against the oracle's model of the same format (numpy, scale by frexp, codes by rint): byte for byte on heavy-tailed, sparse,
    tiny, huge and all-zero blocks; weights (h6 first) are the activation layout with the two FP6 blocks swapped; a forced weight scale
    (the projection folded into another convolution's sums) only moves the exponent."""
    import numpy as np
    from fgvc_amd import ops
    from oracle import fgvc_oracle as O
    g = torch.Generator().manual_seed(3)
    x = torch.randn(400, 32, generator=g).abs() ** 1.5 * (torch.rand(400, 32, generator=g) > 0.4) * torch.exp2(torch.randint(-12, 9, (400, 1), generator=g).float())
    x[5] = 0
    x[6, 1:] = 0
    x[7] = 7.5 * 16 * torch.sign(torch.randn(32, generator=g))
    for s_log2 in (0, 5, -3):
        want = O.act_f16f6_rows(x.numpy(), s_log2)
        xs = x * 2.0 ** s_log2
        h = xs.to(torch.float16)
        got = ops._f16f6_slots(h, xs - h.float(), "l").numpy()
        assert np.array_equal(got, want), int((got != want).sum())
        w_rows = ops._f16f6_slots(h, xs - h.float(), "h").numpy()           # the weight layout: h6 in slots 4 / 6, l6 in 5 / 7
        assert np.array_equal(w_rows[:, :64], want[:, :64])
        assert np.array_equal(w_rows[:, 64:80], want[:, 80:96]) and np.array_equal(w_rows[:, 80:96], want[:, 64:80])
        assert np.array_equal(w_rows[:, 96:112], want[:, 112:128]) and np.array_equal(w_rows[:, 112:128], want[:, 96:112])
        hh, h6, l6 = O.act_f16f6_decode(want)
        top = np.abs(hh).max(1, keepdims=True) + 1e-30
        assert float((np.abs(h6 - hh) / top).max()) <= 1.0 / 15 + 1e-6             # half an e2m3 step of the block's scale: 0.25 x 2^s <= max / 15
    wt = torch.randn(256, 64, 1, 1, generator=g) * 0.1
    bn = torch.nn.BatchNorm2d(256).eval()
    a, ba, ea = ops.prepare_conv_split_f16(wt, bn, ops.ACT_F16F6)
    b, bb, eb = ops.prepare_conv_split_f16(wt, bn, ops.ACT_F16F6, force_exp=ea - 2)
    assert eb == ea - 2 and torch.equal(ba, bb)
    ha = a.view(torch.uint8).reshape(-1, 128)[:, :64].contiguous().view(torch.float16).float()
    hb = b.view(torch.uint8).reshape(-1, 128)[:, :64].contiguous().view(torch.float16).float()
    assert torch.equal(ha, hb * 4.0)                                               # the same weights, two binades lower
    with pytest.raises(AssertionError):
        ops.prepare_conv_split_f16(wt, bn, ops.ACT_F16F6, force_exp=ea + 8)        # would leave the f16 range


def test_stream_kernels_use_no_scratch():
    """What the compiler actually allocated, read from the code-object notes of the built library (tools/kernel_notes.py; LLVM tools only,
    no GPU): every instance of conv256p_kernel -- the one-wave-per-SIMD convolutions whose main loop is one assembly statement that names
    s32 and m0 among its clobbers -- has NO scratch memory (private_segment_fixed_size == 0: no stack, so s32 is no stack pointer) and no
    spilled vector register (round-5 review: the two hottest instances carried 68 / 76 bytes); likewise conv64p_kernel, and the shipped
    instances of both pair kernels (the s_memtime probe instances aside)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("kernel_notes", os.path.join(root, "tools", "kernel_notes.py"))
    kn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kn)
    notes = kn.kernel_notes()
    c256 = {k: v for k, v in notes.items() if "conv256p_kernel" in k}
    assert len(c256) >= 10, sorted(notes)[:5]
    for fam, want in (("conv256p_kernel", 11), ("conv64p_kernel", 3), ("pair_topk_kernel_v7ILi10ELb0", 1), ("pair_topk_kernel_v7ILi5ELb0", 1),
                      ("pair_topk_kernel_v8ILi10ELb0", 1), ("pair_topk_kernel_v8ILi5ELb0", 1)):
        ks = {k: v for k, v in notes.items() if fam in k}
        assert len(ks) >= want, (fam, len(ks))
        for k, v in ks.items():
            assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0, (k, v)
    for k, v in c256.items():                                   # one wave per SIMD: the whole register file, all accumulators in it
        assert v["vgpr_count"] in (376, 512) and v["agpr_count"] in (128, 256), (k, v)


def test_production_library_refuses_the_ablation_switches():
    """fgvc_set_option of libfgvc_hip.so takes the A/B switches, probes and the fault injection, and refuses every bit that makes a kernel
    return wrong results (round-5 review: the production library shipped them); libfgvc_hip_ablations.so, built beside it from the same
    objects, takes them.  Host only."""
    from fgvc_amd import _lib
    lib = _lib.load()
    for name, v in (("pair_debug", 1), ("corr_debug", 1), ("conv_s2_debug", 2), ("corr6_debug", 1024), ("corr8_debug", 2), ("conv_debug", 4),
                    ("pair_f16_debug", 2), ("pair_f16_debug", 4194304 + 1048576), ("pair_f16_debug", 2048)):
        assert lib.fgvc_set_option(name.encode(), v) == _lib.ERR_UNSUPPORTED, (name, v)
        assert b"ablation" in lib.fgvc_last_error()
    for name, v in (("pair_f16_debug", 4096), ("pair_f16_debug", 4194304 + 2048 + 8), ("pair_f16_debug", 512 + 1024), ("conv_debug", 1024 + 16 + 8),
                    ("corr6_debug", 4 + 8), ("conv64_variant", 16), ("readout_prune", 0)):
        assert lib.fgvc_set_option(name.encode(), v) == _lib.FGVC_OK, (name, v)
        assert lib.fgvc_set_option(name.encode(), 1 if name == "readout_prune" else 0) == _lib.FGVC_OK
    with _lib.ablations() as ab:
        assert ab is not lib and ab.fgvc_set_option(b"corr6_debug", 1024) == _lib.FGVC_OK and ab.fgvc_set_option(b"corr6_debug", 0) == _lib.FGVC_OK
    assert _lib.load() is lib
