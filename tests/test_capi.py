"""The C-ABI library builds for gfx950, loads, and exports every symbol include/vidtok_amd.h
declares; argument validation works without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from util import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "vidtok_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(built_lib):
    from vidtok_amd import lib

    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(built_lib, name), f"{name} declared in include/vidtok_amd.h but not exported"
    assert set(lib.SIGNATURES) == set(declared), set(lib.SIGNATURES) ^ set(declared)
    assert built_lib.vt_version() >= 100


def test_conv_desc_layout_matches_header(built_lib):
    from vidtok_amd import lib

    assert built_lib.vt_conv_desc_size() == C.sizeof(lib.ConvDesc)
    assert lib.ConvDesc.xs_z.offset % 8 == 0


def test_conv_desc_field_order_matches_header_and_integration_doc():
    """The ctypes mirror, the header struct and the binding stub shown in INTEGRATION.md list the same fields in the
    same order (a silent drift would shift every field after the edit)."""
    from vidtok_amd import lib

    names = [f[0] for f in lib.ConvDesc._fields_]
    hdr = open(os.path.join(ROOT, "include", "vidtok_amd.h")).read()
    body = hdr[hdr.index("typedef struct vt_conv_desc {") + len("typedef struct vt_conv_desc {"):hdr.index("} vt_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    decl = []
    for stmt in body.split(";"):
        m = re.match(r"\s*(?:const\s+)?(?:void|float|int32_t|int64_t)\s*\*?\s*(.+)$", stmt.strip(), flags=re.S)
        if m:
            decl += [n.strip() for n in m.group(1).split(",")]
    assert decl == names, [(a, b) for a, b in zip(decl, names) if a != b]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blk = doc[doc.index("class vt_conv_desc"):doc.index("assert lib.vt_conv_desc_size()")]
    assert re.findall(r'"(\w+)"', blk) == names


def test_tblock_desc_matches_header_and_integration_doc(built_lib):
    from vidtok_amd import lib

    assert built_lib.vt_tblock_desc_size() == C.sizeof(lib.TBlockDesc)
    names = [f[0] for f in lib.TBlockDesc._fields_]
    hdr = open(os.path.join(ROOT, "include", "vidtok_amd.h")).read()
    body = hdr[hdr.index("typedef struct vt_tblock_desc {") + len("typedef struct vt_tblock_desc {"):hdr.index("} vt_tblock_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    decl = []
    for stmt in body.split(";"):
        m = re.match(r"\s*(?:const\s+)?(?:void|float|int32_t|int64_t)\s*\*?\s*(.+)$", stmt.strip(), flags=re.S)
        if m:
            decl += [n.strip() for n in m.group(1).split(",")]
    assert decl == names, (decl, names)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blk = doc[doc.index("class vt_tblock_desc"):doc.index("lib.vt_temporal_block.argtypes")]
    assert re.findall(r'"(\w+)"', blk) == names
    # unsupported shapes are refused with a message, never launched (no GPU needed to find out)
    d = lib.TBlockDesc()
    d.dtype, d.C, d.ld, d.B, d.T, d.HW, d.tmode = lib.VT_F32, 128, 128, 1, 4, 64, lib.VT_TPAD_ZERO
    assert built_lib.vt_temporal_block_supported(C.byref(d)) == 0
    d.dtype = lib.VT_BF16
    assert built_lib.vt_temporal_block_supported(C.byref(d)) == 1
    d.HW = 60
    assert built_lib.vt_temporal_block_supported(C.byref(d)) == 0
    assert built_lib.vt_temporal_block(C.byref(d), None) != 0 and b"vt_temporal_block" in built_lib.vt_last_error()


def test_argument_validation_without_gpu(built_lib):
    from vidtok_amd import lib

    assert built_lib.vt_conv(None, None) == -1
    assert b"null descriptor" in built_lib.vt_last_error()
    d = lib.ConvDesc()
    d.x, d.w, d.y = 16, 16, 16          # never dereferenced: validation fails first
    d.B = d.Ti = d.Hi = d.Wi = 1
    d.Cin = 3                            # not a multiple of the 16-byte vector
    d.To = d.Ho = d.Wo = d.Cout = 1
    d.KT = d.KH = d.KW = d.st = d.sh = d.sw = 1
    d.ldw = 3
    assert built_lib.vt_conv(C.byref(d), None) == -1
    assert b"Cin" in built_lib.vt_last_error()
    assert built_lib.vt_layernorm_act(16, 0, 104, 16, 0, 104, 16, 16, 10, 100, 1e-6, 1, None) == -1
    assert b"unsupported channel count" in built_lib.vt_last_error()
    # the measurement aids validate like the calls they wrap, before anything is launched
    assert built_lib.vt_conv_profile(C.byref(d), None, None) == -1 and b"null output" in built_lib.vt_last_error()
    assert built_lib.vt_conv_profile(C.byref(d), 16, None) == -1 and b"Cin" in built_lib.vt_last_error()
    t = lib.TBlockDesc()
    assert built_lib.vt_temporal_block_profile(C.byref(t), None, None) == -1 and b"null output" in built_lib.vt_last_error()
    assert built_lib.vt_temporal_block_profile(C.byref(t), 16, None) == -1 and b"only bf16" in built_lib.vt_last_error()
    assert built_lib.vt_temporal_block(C.byref(t), None) == -1


def test_fsq_constants_match_reference_formula(built_lib):
    import torch
    from vidtok_amd import ops

    for levels in ([8, 8, 8, 8, 8], [8, 5, 5, 5], [7, 5, 5, 5, 5], [8] * 6):
        lv = torch.tensor(levels, dtype=torch.int32)
        half_l = (lv - 1) * (1 + 1e-3) / 2                      # reference regularizers.py:155-157
        offset = torch.where(lv % 2 == 0, 0.5, 0.0)
        shift = (offset / half_l).atanh()
        h, o, s, b = ops.fsq_consts(levels)
        assert h == half_l.tolist() and o == offset.tolist() and s == shift.tolist()
        assert b == [float(x) for x in torch.cumprod(torch.tensor([1] + levels[:-1]), 0)]


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from vidtok_amd import lib

    monkeypatch.setattr(lib, "_lib", None)
    with pytest.raises(lib.VtError):
        lib.load(str(tmp_path / "nope.so"))


def test_conv_plan_tile_selection(built_lib, monkeypatch):
    """vt_conv_plan (no GPU): the 8-wave 256x256 tile is what BASELINE-sized Cout % 256 == 0 layers get, the conv_tile option
    forces / forbids it for small parity cases (option conv_tile), LayerNorm is fused only for Cout = 128 on full tiles."""
    from vidtok_amd import lib as L
    from vidtok_amd import ops

    def desc(M_hw, cin, cout, **kw):
        d = L.ConvDesc()
        d.x = d.w = d.y = 4096
        d.B, d.Ti, d.Hi, d.Wi, d.Cin = 1, 1, M_hw[0], M_hw[1], cin
        d.To, d.Ho, d.Wo, d.Cout = 1, M_hw[0], M_hw[1], cout
        d.ldw, d.ldy = 9 * cin, cout
        d.KT, d.KH, d.KW = 1, 3, 3
        d.st = d.sh = d.sw = 1
        d.ph = d.pw = 1
        d.dtype = d.out_dtype = L.VT_BF16
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    L.load().vt_reset_options()
    big = ops.conv_plan(desc((1280, 1024), 256, 256))            # the 256-channel level of the benchmark (B=4: 20 frames), K = 2 304
    assert big["tile"] == (256, 256) and big["waves"] == 8 and big["workgroups"] == 5120 and big["lds_epilogue"]
    f16 = ops.conv_plan(desc((1280, 1024), 256, 256, dtype=L.VT_F16, out_dtype=L.VT_F16))      # fp16: every decision of bf16
    assert f16 == big
    assert not ops.conv_plan(desc((1280, 1024), 256, 256, dtype=L.VT_F32, out_dtype=L.VT_F32))["lds_epilogue"]   # plain fp32 rows: the vector epilogue
    small = ops.conv_plan(desc((64, 64), 256, 256))
    assert small["tile"] == (128, 128) and small["deep_ring"] and not small["lds_epilogue"]      # 64 tiles <= CUs: 4-slot ring
    with L.options(conv_deep=0):
        assert not ops.conv_plan(desc((64, 64), 256, 256))["deep_ring"]
    with L.options(conv_tile=128):
        assert not ops.conv_plan(desc((1280, 1024), 256, 256))["deep_ring"]                      # 20 480 tiles: two workgroups per CU
    assert ops.conv_plan(desc((64, 64), 128, 8))["tile"] == (256, 32)
    assert ops.conv_plan(desc((64, 64), 128, 64))["tile"] == (256, 64)
    with L.options(conv_tile=256):
        assert ops.conv_plan(desc((64, 64), 256, 256))["tile"] == (256, 256)
        assert ops.conv_plan(desc((64, 64), 256, 128))["tile"] == (128, 128)     # not legal there
    with L.options(conv_tile=128):
        assert ops.conv_plan(desc((1280, 1024), 256, 256))["tile"] == (128, 128)
    assert L.get_option("conv_tile") == 0
    ln = dict(ln_mode=2, ln_gamma=4096, ln_beta=4096, ln_out=4096, ldn=128, ln_eps=1e-6)
    p = ops.conv_plan(desc((64, 64), 128, 128, **ln))
    assert p["ln_fused"] and p["launches"] == 1
    ln["ldn"] = 256
    p = ops.conv_plan(desc((64, 64), 128, 256, **ln))
    assert not p["ln_fused"] and p["launches"] == 2
    # the weight-stationary persistent kernel: bf16, 3x3, Cin = Cout = 128, frames tiling by 8 x 16
    p = ops.conv_plan(desc((256, 256), 128, 128))
    assert p["kernel"] == "ws2" and p["tile"] == (64, 128) and p["waves"] == 8 and p["workgroups"] == 64 * 16      # conv_ws2.hip: 4 x 16-pixel tiles
    assert ops.conv_plan(desc((256, 256), 128, 128, dtype=L.VT_F16, out_dtype=L.VT_F16)) == p
    assert ops.conv_plan(desc((256, 256), 128, 128, **dict(ln, ldn=128)))["ln_fused"]
    assert ops.conv_plan(desc((256, 250), 128, 128))["kernel"] == "igemm"
    assert ops.conv_plan(desc((256, 256), 128, 128, dtype=L.VT_F32, out_dtype=L.VT_F32))["kernel"] == "igemm"
    with L.options(conv_ws=0):
        assert ops.conv_plan(desc((256, 256), 128, 128))["kernel"] == "igemm"
    # the narrow-output kernel: the decoder's conv_out at the benchmark size (B = 4, 20 frames, 3 trimmed, 256 x 256)
    co = dict(KT=3, pt=2, ldw=27 * 128, out_dtype=L.VT_F32, out_layout=L.VT_NCTHW, t_trim=3, B=4, Ti=20, To=20)
    p = ops.conv_plan(desc((256, 256), 128, 3, **co))
    assert p["kernel"] == "narrow" and p["tile"] == (8 * 14, 16) and p["launches"] == 1
    assert p["workgroups"] == 4 * 2 * 32 * 5                     # clips x time segments x row blocks x groups of 4 windows
    assert ops.conv_plan(desc((256, 256), 128, 3, **dict(co, out_layout=L.VT_NDHWC, t_trim=0, ldy=4)))["kernel"] == "igemm"
    assert ops.conv_plan(desc((256, 256), 128, 8, **co))["kernel"] == "igemm"
    with L.options(conv_narrow=0):
        assert ops.conv_plan(desc((256, 256), 128, 3, **co))["kernel"] == "igemm"
    with pytest.raises(L.VtError):
        ops.conv_plan(desc((64, 64), 100, 128))                  # same validation as vt_conv
    # split-K over the tap planes (vt_conv_work_bytes; no GPU: 256 CUs assumed): on by default since round 5 -- the decision looks at
    # ONE clip's pixels (To * Ho * Wo), never at B, so a clip's bits do not depend on its batch (tests/test_gpu_ops.py::
    # test_conv_split_k_does_not_depend_on_the_batch, tests/test_gpu_e2e.py::test_full_size_properties)
    import ctypes as C
    lib = L.load()
    deep = desc((64, 64), 512, 512, work=4096)                  # 4 096 pixels per clip, K = 4 608: 128 tiles of 128 x 128 -- every workgroup alone on its CU
    assert L.get_option("conv_splitk") == 1
    assert lib.vt_conv_work_bytes(C.byref(deep)) == 3 * 4096 * 512 * 4 and ops.conv_plan(deep)["launches"] == 2     # the three rows of the 3 x 3 + the reduction
    assert lib.vt_conv_work_bytes(C.byref(desc((64, 64), 512, 512, B=8))) == 8 * 3 * 4096 * 512 * 4                # a batch of such clips splits too
    assert lib.vt_conv_work_bytes(C.byref(desc((256, 256), 512, 512))) == 0                                        # 65 536 pixels per clip: enough tiles
    assert lib.vt_conv_work_bytes(C.byref(desc((64, 64), 512, 512, dtype=L.VT_F32, out_dtype=L.VT_F32))) == 0      # the 16-bit types only
    assert lib.vt_conv_work_bytes(C.byref(desc((64, 64), 512, 512, work=4096, dtype=L.VT_F16, out_dtype=L.VT_F16))) == 3 * 4096 * 512 * 4
    assert lib.vt_conv_work_bytes(C.byref(desc((64, 64), 128, 512, ldw=9 * 128))) == 0                              # K = 1 152: too short to pay
    with L.options(conv_splitk=0):
        assert lib.vt_conv_work_bytes(C.byref(deep)) == 0 and ops.conv_plan(deep)["launches"] == 1


def test_options_table(built_lib, monkeypatch):
    """vt_set_option / vt_get_option / vt_reset_options: every option is listed in the header's comment, unknown names are
    errors, the environment only provides the defaults (read by vt_reset_options, never by a launch)."""
    from vidtok_amd import lib as L

    names = L.option_names()
    hdr = open(os.path.join(ROOT, "include", "vidtok_amd.h")).read()
    doc = hdr[hdr.index("Process-wide tuning / test switches"):hdr.index("int vt_set_option")]
    assert len(names) == built_lib.vt_option_count() >= 12 and built_lib.vt_option_name(len(names)) is None
    for n in names:
        assert re.search(r"\b" + n + r"\b", doc), f"option {n} is not documented in include/vidtok_amd.h"
    with pytest.raises(L.VtError):
        L.set_option("no_such_option", 1)
    with pytest.raises(L.VtError):
        L.get_option("no_such_option")
    built_lib.vt_reset_options()
    assert L.get_option("conv_ws") == 2 and L.get_option("conv_tile_min") == 128
    with L.options(conv_ws=0, conv_tile_min=7):
        assert L.get_option("conv_ws") == 0 and L.get_option("conv_tile_min") == 7
        monkeypatch.setenv("VT_CONV_WS", "1")          # the environment is not consulted after start-up ...
        assert L.get_option("conv_ws") == 0
    assert L.get_option("conv_ws") == 2
    built_lib.vt_reset_options()                           # ... except by an explicit reset
    assert L.get_option("conv_ws") == 1
    monkeypatch.delenv("VT_CONV_WS")
    built_lib.vt_reset_options()
    assert L.get_option("conv_ws") == 2


def test_bench_committed_traffic_fallback(tmp_path, monkeypatch):
    """bench.py's roofline.traffic fallback reads every committed PMC pass under profiles/ (the r02
    file spells the key `bytes_per_launch`, r01 `traffic_bytes_per_launch`; a KeyError there cost the whole JSON line)."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    v, src = bench.committed_traffic("bf16", 4)
    newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+_conv_traffic_pmc\.json", f))[-1]
    rec = json.load(open(os.path.join(ROOT, "profiles", newest)))
    assert v == round(rec.get("bytes_per_launch", rec.get("traffic_bytes_per_launch"))) and newest in src and v > 1e8
    v32, src32 = bench.committed_traffic("fp32", 4)
    assert v32 is not None and v32 > v and "fp32" in src32
    assert bench.committed_traffic("bf16", 8)[0] is None                     # the committed passes are B=4 runs
    # every committed file parses under one of the two spellings
    for f in os.listdir(os.path.join(ROOT, "profiles")):
        if re.fullmatch(r"r\d+_conv_traffic_pmc(_fp32)?\.json", f):
            r = json.load(open(os.path.join(ROOT, "profiles", f)))
            assert "bytes_per_launch" in r or "traffic_bytes_per_launch" in r, f
    # synthetic profiles/: a broken newest file and an unknown spelling are skipped, the older spelling is found
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r03_conv_traffic_pmc.json").write_text("{not json")
    (prof / "r02_conv_traffic_pmc.json").write_text(json.dumps({"something_else": 1}))
    (prof / "r01_conv_traffic_pmc.json").write_text(json.dumps({"traffic_bytes_per_launch": 123.4}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.committed_traffic("bf16", 4) == (123, "profiles/r01_conv_traffic_pmc.json (committed pass)")
    assert bench.committed_traffic("fp32", 4)[0] is None


def test_model_handle_without_gpu(built_lib):
    """The handle-level C-ABI (csrc/model.cpp) up to the first launch: config validation, the struct mirror, the list of
    reference state_dict keys the C++ stage graph reads (= the encoder / decoder keys of the Python model), latent
    dimensions and workspace sizing -- all of it host-side (a dry run of the graph), no GPU needed."""
    import vidtok_amd
    from vidtok_amd import lib

    assert built_lib.vt_model_config_size() == C.sizeof(lib.ModelConfig)
    hdr = open(os.path.join(ROOT, "include", "vidtok_amd.h")).read()
    body = hdr[hdr.index("typedef struct vt_model_config {") + len("typedef struct vt_model_config {"):hdr.index("} vt_model_config;")]
    names = re.findall(r"\b([a-z_0-9]+)(?:\[8\])?\s*[,;]", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert names == [f[0] for f in lib.ModelConfig._fields_], names
    cfg = vidtok_amd.load_config(os.path.join(ROOT, "configs", "vidtok_kl_causal_488_4chn.yaml"))
    enc = cfg["model"]["params"]["encoder_config"]["params"]
    mc = lib.ModelConfig()
    mc.ch, mc.num_res_blocks, mc.in_channels, mc.out_ch, mc.z_channels, mc.double_z = enc["ch"], enc["num_res_blocks"], 3, 3, enc["z_channels"], 1
    mc.num_resolutions, mc.time_downsample_factor = 4, 4
    for i, v in enumerate(enc["ch_mult"]):
        mc.ch_mult[i] = v
    for k, v in dict(spatial_ds=[0, 1, 2], tempo_ds=[2, 1], spatial_us=[1, 2, 3], tempo_us=[1, 2]).items():
        setattr(mc, "n_" + k, len(v))
        for i, e in enumerate(v):
            getattr(mc, k)[i] = e
    h = C.c_void_p()
    mc.version = 3
    assert built_lib.vt_create(C.byref(mc), lib.VT_BF16, C.byref(h)) != 0 and b"version" in built_lib.vt_last_error()
    mc.version, mc.interpolation_mode = 0, 1
    assert built_lib.vt_create(C.byref(mc), lib.VT_BF16, C.byref(h)) != 0 and b"interpolation_mode" in built_lib.vt_last_error()
    # v1.1: same parameters, a front pad up to a multiple of 4 instead of 3 frames, one pass per clip
    mc.version = 1
    assert built_lib.vt_create(C.byref(mc), lib.VT_BF16, C.byref(h)) == 0
    try:
        cfg11 = vidtok_amd.load_config(os.path.join(ROOT, "configs", "vidtok_v1_1", "vidtok_kl_causal_488_4chn_v1_1.yaml"))
        sd11 = vidtok_amd.load_model_from_config(cfg11, verbose=False).state_dict()
        assert {built_lib.vt_weight_name(h, i).decode() for i in range(built_lib.vt_weight_count(h))} == set(sd11.keys())
        ld = (C.c_int32 * 4)()
        assert built_lib.vt_latent_dims(h, 17, 256, 256, ld) == 0 and list(ld) == [8, 5, 32, 32]      # 17 + 3
        assert built_lib.vt_latent_dims(h, 18, 256, 256, ld) == 0 and list(ld) == [8, 5, 32, 32]      # 18 + 2 (v1.0: 18 + 3 -> 5 too)
        assert built_lib.vt_latent_dims(h, 21, 256, 256, ld) == 0 and list(ld) == [8, 6, 32, 32]
        assert built_lib.vt_workspace_bytes(h, 1, 18, 64, 64) > 0
    finally:
        built_lib.vt_destroy(h)
    mc.version, mc.interpolation_mode = 0, 0
    assert built_lib.vt_create(C.byref(mc), 7, C.byref(h)) != 0 and b"dtype" in built_lib.vt_last_error()
    assert built_lib.vt_create(C.byref(mc), lib.VT_BF16, C.byref(h)) == 0
    try:
        keys = [built_lib.vt_weight_name(h, i).decode() for i in range(built_lib.vt_weight_count(h))]
        model = vidtok_amd.load_model_from_config(cfg, verbose=False)
        sd = model.state_dict()
        assert set(keys) == set(sd.keys()) and len(keys) == len(set(keys))
        for i, k in enumerate(keys):           # ... with the reference's parameter shapes
            shp, nd = (C.c_int64 * 5)(), C.c_int32()
            assert built_lib.vt_weight_shape(h, i, shp, C.byref(nd)) == 0 and tuple(shp[:nd.value]) == tuple(sd[k].shape), k
        import torch
        w = torch.zeros(sd[keys[0]].shape)
        bad = (C.c_int64 * 1)(3)
        assert built_lib.vt_load_weight(h, keys[0].encode(), w.data_ptr(), bad, 1) != 0 and b"wrong shape" in built_lib.vt_last_error()
        assert built_lib.vt_load_weight(h, b"loss.logvar", w.data_ptr(), bad, 1) != 0 and b"not a parameter" in built_lib.vt_last_error()
        ld = (C.c_int32 * 4)()
        assert built_lib.vt_latent_dims(h, 17, 256, 256, ld) == 0 and list(ld) == [8, 5, 32, 32]
        small, big = built_lib.vt_workspace_bytes(h, 1, 17, 64, 64), built_lib.vt_workspace_bytes(h, 4, 17, 256, 256)
        assert 0 < small < big < 64 << 30          # B=4 17x256x256 bf16: two arenas of a few activations each
        # nothing was loaded: the first use of a weight says which
        assert built_lib.vt_encode(h, 256, 1, 17, 64, 64, 256, 256, small, None) != 0 and b"was not loaded" in built_lib.vt_last_error()
    finally:
        built_lib.vt_destroy(h)


def _causal_configs():
    import glob

    paths = sorted(glob.glob(os.path.join(ROOT, "configs", "*.yaml")) + glob.glob(os.path.join(ROOT, "configs", "vidtok_v1_1", "*.yaml")))
    return [os.path.relpath(q, os.path.join(ROOT, "configs"))[:-5] for q in paths]


@pytest.mark.parametrize("name", _causal_configs())
def test_model_handle_graph_of_every_causal_config(built_lib, name):
    """The C++ stage graph (csrc/model.cpp) for every shipped YAML, causal v1.0 / v1.1 and non-causal -- every compression schedule
    (4x8x8, 4x16x16, 2x8x8, 4x4x4, 8x8x8), every latent width, both regularizers: the handle lists exactly the encoder /
    decoder parameters of the Python model with the reference's shapes, its latent dimensions follow the schedule, and
    the dry run of both graphs sizes a workspace.  Host-side only."""
    import vidtok_amd
    from util import handle_config
    from vidtok_amd import lib

    cfg = vidtok_amd.load_config(os.path.join(ROOT, "configs", name + ".yaml"))
    prm = cfg["model"]["params"]
    enc = prm["encoder_config"]["params"]
    reg = prm["regularizer_config"]
    mc = handle_config(lib, enc, reg["target"], reg.get("params", {}), prm["encoder_config"]["target"])
    assert mc.version == (1 if "v1_1" in name else (2 if "noncausal" in name else 0))
    h = C.c_void_p()
    assert built_lib.vt_create(C.byref(mc), lib.VT_BF16, C.byref(h)) == 0, built_lib.vt_last_error()
    try:
        sd = {k: v for k, v in vidtok_amd.load_model_from_config(cfg, verbose=False).state_dict().items() if not k.startswith("regularization")}
        n = built_lib.vt_weight_count(h)
        keys = [built_lib.vt_weight_name(h, i).decode() for i in range(n)]
        assert set(keys) == set(sd) and len(keys) == len(sd)
        for i, k in enumerate(keys):
            shp, nd = (C.c_int64 * 5)(), C.c_int32()
            assert built_lib.vt_weight_shape(h, i, shp, C.byref(nd)) == 0 and tuple(shp[:nd.value]) == tuple(sd[k].shape), k
        f = enc.get("time_downsample_factor", 4)
        sds = len(enc.get("spatial_ds") or range(len(enc["ch_mult"]) - 1))
        for T in ((f, 2 * f, 3 * f) if mc.version == 2 else (1, f, f + 1, 2 * f + 1, 3 * f)):
            pad = 0 if T % f == 0 else (f - T % f if mc.version == 1 else f - 1)
            ld = (C.c_int32 * 4)()
            assert built_lib.vt_latent_dims(h, T, 64, 64, ld) == 0
            assert list(ld) == [enc["z_channels"] * (2 if enc.get("double_z", True) else 1), (T + pad) // f, 64 >> sds, 64 >> sds], (T, list(ld))
        assert built_lib.vt_workspace_bytes(h, 1, f + 1, 64, 64) > 0, built_lib.vt_last_error()
    finally:
        built_lib.vt_destroy(h)


@pytest.mark.parametrize("name,ov,reg", [
    ("vidtok_kl_causal_488_4chn", dict(norm_type="groupnorm"), None),
    ("vidtok_v1_1/vidtok_fsq_causal_888_32768_v1_1", dict(norm_type="groupnorm"), None),
    ("vidtok_kl_noncausal_488_4chn", dict(norm_type="groupnorm"), None),
    ("vidtok_fsq_causal_488_32768", dict(z_channels=6), dict(levels=[8, 5, 5], num_codebooks=2)),
    ("vidtok_fsq_causal_488_32768", dict(z_channels=8), dict(levels=[8, 5, 5], num_codebooks=2, dim=8)),
    ("vidtok_fsq_causal_488_32768", dict(z_channels=7), dict(levels=[8, 5, 5], dim=7)),
], ids=["groupnorm_v10", "groupnorm_v11", "groupnorm_noncausal", "fsq_two_codebooks", "fsq_two_codebooks_projected", "fsq_projected"])
def test_model_handle_constructor_variants_without_gpu(built_lib, name, ov, reg):
    """vt_model_config.norm_type / fsq_num_codebooks / fsq_dim: vt_create drives the constructor arguments no shipped
    YAML sets -- GroupNorm parameter keys without the ".norm" level (model_3dcausal.py:30-34), FSQ's project_in / project_out
    (regularizers.py:137-139) -- and lists exactly the tensors of the Python model built with the same overrides; inconsistent
    FSQ widths are refused.  Host-side only (bit-equality with the engine: tests/test_gpu_e2e.py::test_model_handle_constructor_variants)."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import build_model, handle_config
    from vidtok_amd import lib

    model, cfg, sd = build_model(name, overrides=ov, reg_overrides=reg)
    prm = cfg["model"]["params"]
    mc = handle_config(lib, prm["encoder_config"]["params"], prm["regularizer_config"]["target"], prm["regularizer_config"].get("params", {}),
                       prm["encoder_config"]["target"])
    assert mc.norm_type == (1 if "norm_type" in ov else 0)
    h = C.c_void_p()
    assert built_lib.vt_create(C.byref(mc), lib.VT_F32, C.byref(h)) == 0, built_lib.vt_last_error()
    try:
        keys = [built_lib.vt_weight_name(h, i).decode() for i in range(built_lib.vt_weight_count(h))]
        want = {k: v for k, v in sd.items() if not k.startswith("regularization") or ".project_" in k}
        assert set(keys) == set(want) and len(keys) == len(want)
        for i, k in enumerate(keys):
            shp, nd = (C.c_int64 * 5)(), C.c_int32()
            assert built_lib.vt_weight_shape(h, i, shp, C.byref(nd)) == 0 and tuple(shp[:nd.value]) == tuple(want[k].shape), k
        if "norm_type" in ov:
            assert any(k.endswith("norm1.weight") for k in keys) and not any("norm1.norm." in k or ".norm.norm." in k for k in keys)
        assert built_lib.vt_workspace_bytes(h, 1, 8, 64, 64) > 0, built_lib.vt_last_error()
    finally:
        built_lib.vt_destroy(h)
    if reg is not None:
        mc.z_channels += 1                          # z_channels must be FSQ's dim
        assert built_lib.vt_create(C.byref(mc), lib.VT_F32, C.byref(h)) != 0 and b"z_channels" in built_lib.vt_last_error()


def _build_c_example(tmp_path):
    """examples/roundtrip.c with the system C compiler (not hipcc): the boundary is C"""
    import shutil
    import subprocess

    cc = shutil.which("cc") or shutil.which("gcc")
    assert cc, "no C compiler"
    exe = os.path.join(str(tmp_path), "roundtrip")
    libdir = os.path.join(ROOT, "vidtok_amd")
    cmd = [cc, "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "examples", "roundtrip.c"), "-o", exe, "-L", libdir, "-lvidtok_amd", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_example_builds(built_lib, tmp_path):
    assert os.path.exists(_build_c_example(tmp_path))


@pytest.mark.gpu
def test_c_example_runs(built_lib, tmp_path):
    """the plain-C host of examples/roundtrip.c: create, load 416 tensors by reference key, encode -> KL mode -> decode, twice
    on one workspace; exit code 0 = finite reconstruction"""
    import subprocess

    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe, "9", "64", "64"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "finite 1" in r.stdout, (r.stdout, r.stderr)
    # ... and the v1.1 model with temporal tiling from C (chunks of 16 frames, decoder look-ahead: BASELINE.json configs[4]'s
    # protocol on a short clip), next to the one-pass result of the same clip
    r = subprocess.run([exe, "41", "64", "64", "16"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "finite 1" in r.stdout and "tiled (t_chunk_enc 16" in r.stdout, (r.stdout, r.stderr)
