"""CPU: the C-ABI library loads, exports every symbol include/d2s.h declares, and its pure-host
entry points (shape logic, error plumbing) behave like the reference's integer logic.  No compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from desktop2stereo_amd import _lib
from desktop2stereo_amd.config import engine_shape
from oracle import d2s_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from desktop2stereo_amd import build
        build.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(REPO, "include", "d2s.h")).read()
    declared = set(re.findall(r"\b(d2s_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.d2s_version() >= 110          # 110: d2s_dibr_params.struct_size


def test_struct_layout_matches_header():
    assert C.sizeof(_lib.ModelDesc) == 4 * (3 + 4 + 4 + 5) + 4 + 4 + 4 + 4  # ... ln_eps, precision, temporal, max_depth
    assert C.sizeof(_lib.PostParams) == 28 and _lib.PostParams.metric.offset == 24
    assert C.sizeof(_lib.SbsParams) == 24 and _lib.SbsParams.ipd_uv.offset == 0 and _lib.SbsParams.depth_ratio.offset == 8
    assert (C.sizeof(_lib.DibrParams) == 80 and _lib.DibrParams.corner_radius.offset == 52 and _lib.DibrParams.viewport.offset == 56
            and _lib.DibrParams.alpha_mode.offset == 72 and _lib.DibrParams.struct_size.offset == 76)   # struct_size sits in what was tail padding
    assert C.sizeof(_lib.PreParams) == 32 and _lib.PreParams.std.offset == 12 and _lib.PreParams.resample.offset == 24 and _lib.PreParams.square.offset == 28


def test_sbs_shape_matches_reference_padding(lib):
    from desktop2stereo_amd.ops import sbs_params, sbs_shape
    for (H, W) in [(1080, 1920), (90, 160), (120, 160), (100, 240), (75, 133), (2160, 3840), (480, 640), (1, 1)]:
        for mode in _lib.MODE:
            for fill in (False, True):
                e = np.zeros((1, H, W), np.float32)
                if fill:
                    e = O.pad_to_aspect(e)
                hp, wp = e.shape[1:]
                want = {"Half-SBS": (hp, wp), "Half-TAB": (hp, wp), "Full-SBS": (hp, 2 * wp), "Full-TAB": (2 * hp, wp)}[mode]
                assert sbs_shape(H, W, sbs_params(display_mode=mode, fill_16_9=fill)) == want, (H, W, mode, fill)


def test_jpeg_bound_host_logic(lib):
    """d2s_jpeg_bound is pure host arithmetic: 16x16 MCUs, 6 blocks each, <= 216 bytes per block before stuffing."""
    from desktop2stereo_amd.ops import jpeg_bound
    for (H, W) in [(1080, 3840), (1, 1), (17, 33), (4320, 3840)]:
        out_b, ws_b = jpeg_bound(H, W)
        nmcu = -(-H // 16) * -(-W // 16)
        assert out_b >= 623 + 2 + 2 * nmcu * 6 * 216                     # header + EOI + every stream byte stuffed
        assert ws_b >= nmcu * 6 * 64 * 2 + nmcu * 6 * 216 and ws_b % 256 == 0
    ob, wb = C.c_int64(), C.c_int64()
    assert lib.d2s_jpeg_bound(0, 10, C.byref(ob), C.byref(wb)) != 0 and lib.d2s_last_error()
    assert lib.d2s_jpeg_bound(70000, 10, C.byref(ob), C.byref(wb)) != 0                  # JPEG dimensions are 16-bit


def test_errors_are_loud(lib):
    sp = _lib.SbsParams(0.064, 2.0, 0.0, 7, 0)
    oh, ow = C.c_int(), C.c_int()
    rc = lib.d2s_sbs_shape(10, 10, C.byref(sp), C.byref(oh), C.byref(ow))
    assert rc != 0 and lib.d2s_last_error()
    with pytest.raises(_lib.D2SError):
        _lib.check(rc, "d2s_sbs_shape")
    with pytest.raises(_lib.D2SError):
        _lib.load.__wrapped__ if hasattr(_lib.load, "__wrapped__") else None
        _lib._lib = None
        try:
            _lib.load("/nonexistent/libd2s_hip.so")
        finally:
            _lib._lib = None
            _lib.load()


def test_engine_shape_host_logic():
    # same table as the oracle's (reference depth.py:676-706)
    for args in [(1080, 1920, 518), (2160, 3840, 518), (1440, 2560, 518), (720, 1280, 518), (1080, 1920, 336), (90, 160, 84),
                 (600, 800, 518), (1200, 1200, 392), (333, 777, 238)]:
        assert engine_shape(*args) == O.engine_shape(*args)


def test_weight_generator_matches_hf_layout():
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.weights import expected_shapes, make_weights
    for name, n_params in [("vits", 24.8e6), ("vitb", 97.5e6), ("vitl", 335.3e6)]:
        shapes = expected_shapes(MODELS[name])
        total = sum(int(np.prod(s)) for s in shapes.values())
        # + mask_token (D) which HF holds but the engine never reads
        assert abs(total + MODELS[name].hidden - n_params) / n_params < 0.005, (name, total)
    w1, w2 = make_weights(MODELS["tiny"], 0), make_weights(MODELS["tiny"], 0)
    assert all(np.array_equal(w1[k], w2[k]) for k in w1)
    assert not np.array_equal(make_weights(MODELS["tiny"], 1)["head.conv1.weight"], w1["head.conv1.weight"])


def test_load_safetensors_from_hf_checkpoint(tmp_path):
    """weights.load_safetensors on a file written by HF save_pretrained(safe_serialization=True) -- the format the
    reference's convert.py:14-24 produces and depth.py:1649-1662 loads -- returns exactly the model's tensors."""
    torch = pytest.importorskip("torch")
    pytest.importorskip("transformers")
    from transformers import DepthAnythingConfig, DepthAnythingForDepthEstimation
    from desktop2stereo_amd.config import MODELS
    from desktop2stereo_amd.weights import expected_shapes, load_safetensors
    cfg = MODELS["tiny"]
    hf = DepthAnythingConfig(
        backbone_config=dict(model_type="dinov2", hidden_size=cfg.hidden, num_attention_heads=cfg.heads,
                             num_hidden_layers=cfg.layers, image_size=518, patch_size=14, out_indices=list(cfg.out_indices),
                             apply_layernorm=True, reshape_hidden_states=False),
        reassemble_hidden_size=cfg.hidden, neck_hidden_sizes=list(cfg.neck), fusion_hidden_size=cfg.fusion,
        head_hidden_size=cfg.head_hidden)
    torch.manual_seed(1)
    m = DepthAnythingForDepthEstimation(hf)
    m.half().save_pretrained(str(tmp_path), safe_serialization=True)            # fp16 on disk, like the published checkpoints
    got = load_safetensors(os.path.join(str(tmp_path), "model.safetensors"), cfg)
    sd = m.state_dict()
    assert set(got) == set(expected_shapes(cfg))
    for k, v in got.items():
        assert v.dtype == np.float32 and np.array_equal(v, sd[k].float().numpy()), k
    with pytest.raises((KeyError, ValueError)):
        load_safetensors(os.path.join(str(tmp_path), "model.safetensors"), MODELS["vits"])      # wrong architecture: loud


def test_bench_tile_fit_batch_and_traffic_table():
    """bench.py host logic: the tile-fitting batch (whole rounds of 256 x 256 tiles on 256 CUs) and the committed PMC
    traffic table the roofline objects quote."""
    import json
    import bench
    # ViT-B at 294 x 518: 778 tokens; 27 frames -> 83 tile rows: 249 / 747 / 996 tiles = whole rounds
    assert bench.tile_fit_batch(778, 768, 3072, 32) == 27
    b = bench.tile_fit_batch(778, 768, 3072, 16)
    assert 8 < b <= 16
    for tokens, hidden in ((778, 384), (778, 1024), (337, 768)):
        b = bench.tile_fit_batch(tokens, hidden, 4 * hidden, 32)
        assert 16 < b <= 32
    # the PMC table is quoted only while it describes THIS tree (round 5): its kernel-source digest must equal the tree's
    doc, src = bench.pmc_traffic_file()
    t = bench.pmc_traffic("gemm_linear", 1, True)
    if src["status"].startswith("ok"):
        assert doc is not None and t is not None and 5e6 < t < 5e7          # ~15.7 MB per launch at batch 1
    else:
        assert t is None and src["status"].startswith(("stale", "missing")), src
    from desktop2stereo_amd.build import kernel_sources_digest
    assert len(kernel_sources_digest()) == 64 and kernel_sources_digest() == kernel_sources_digest()
    assert bench.pmc_traffic("gemm_linear", 1, False) is None   # only for the workload the profile was taken on
    assert bench.pmc_traffic("no_such_class", 1, True) is None
    with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
        tab = json.load(f)
    assert {"gemm_linear", "stereo_warp"} <= set(tab["traffic_bytes_per_launch"])


def test_bench_compact_line_is_small_and_complete():
    """The driver parses the LAST stdout line of bench.py and keeps an ~8 KB tail (round 5's single 21 KB line left BENCH_r05.parsed null).
    bench.compact_line() applied to a full report committed under profiles/ must stay below 6 000 bytes and carry the contract's keys,
    the roofline (with traffic and its provenance) and the CPU baseline -- checked here without a GPU."""
    import importlib.util
    import json
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("d2s_bench_for_test", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import glob
    full_path = sorted(glob.glob(os.path.join(repo, "profiles", "r*_bench_driver_cmd_full.json")))[-1]      # the latest evidence round's report
    with open(full_path) as f:
        full = json.load(f)
    c = bench.compact_line(full, os.path.join(repo, "gpurun_out", "bench_full.json"))
    line = json.dumps(c, separators=(",", ":"))
    assert len(line.encode()) < 6000, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "roofline_warp", "cpu_baseline", "batched", "depth_l1_vs_ref", "warp_max_lsb", "full_report"):
        assert k in c, k
    assert set(c["config"]) >= {"workload", "frames_per_step_per_gpu", "timed_region"} and "model" not in c["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"], k
    assert abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-4
    assert c["cpu_baseline"]["kind"] in ("port", "reference") and c["cpu_baseline"]["cores"] >= 1 and "sample" in c["cpu_baseline"]
    assert abs(c["value"] - full["value"]) <= 1e-4 * full["value"]
    # a report ten times as wordy must still fit (the line is built from picked keys, not from the report's size)
    fat = dict(full, config=dict(full["config"], workload=full["config"]["workload"] * 10))
    assert len(json.dumps(bench.compact_line(fat, "x"), separators=(",", ":")).encode()) < 6000
