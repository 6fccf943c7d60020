"""CPU: oracle/jpeg_oracle.py against the libjpeg-turbo golden vectors (tests/golden/jpeg_kat.*,
made by tests/golden/make_jpeg_golden.py) — this is what pins the MJPEG-sink oracle."""
import importlib.util
import json
import os

import numpy as np
import pytest

from oracle import jpeg_oracle as J


def _cases(golden_dir):
    spec = importlib.util.spec_from_file_location("make_jpeg_golden", os.path.join(golden_dir, "make_jpeg_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    z = np.load(os.path.join(golden_dir, "jpeg_kat.npz"))
    with open(os.path.join(golden_dir, "jpeg_kat.json")) as f:
        meta = json.load(f)
    for i, c in enumerate(meta["cases"]):
        yield c, mod.jpeg_case_input(c["H"], c["W"], c["kind"], c["seed"]), z[f"jpeg_{i}"].tobytes()


def test_oracle_matches_libjpeg_turbo_bytes(golden_dir):
    n = 0
    for c, rgb, want in _cases(golden_dir):
        got = J.encode_jpeg(rgb, c["quality"])
        assert got == want, f"{c}: oracle stream differs from libjpeg-turbo's ({len(got)} vs {len(want)} bytes)"
        n += 1
    assert n >= 12


def test_float_frames_round_half_even(golden_dir):
    # make_sbs returns float32 0..255; cv2.imencode converts with convertTo(CV_8U): rint + saturate
    f = np.array([[[0.5, 1.5, 2.5], [254.5, 255.5, -3.0]], [[127.49, 127.51, 300.0], [10.0, 20.0, 30.0]]], dtype=np.float32)
    u = J.to_u8(f)
    assert u.tolist() == [[[0, 2, 2], [254, 255, 0]], [[127, 128, 255], [10, 20, 30]]]
    assert J.encode_jpeg(f, 90) == J.encode_jpeg(u, 90)


def test_quality_tables():
    ql, qc = J.quant_tables(100)
    assert ql.min() == 1 and ql.max() == 1 and qc.max() == 1
    ql, qc = J.quant_tables(50)
    assert (ql == J.STD_LUMA_Q).all() and (qc == J.STD_CHROMA_Q).all()
    ql, _ = J.quant_tables(1)
    assert ql.max() == 255                                             # force_baseline clamp


def test_pillow_agrees_when_present(golden_dir):
    """Live re-check of the fixture premise where Pillow exists (it does in this image)."""
    PIL = pytest.importorskip("PIL.Image")
    import io
    rgb = np.random.default_rng(5).integers(0, 256, (40, 56, 3), dtype=np.uint8)
    buf = io.BytesIO()
    PIL.fromarray(rgb).save(buf, "JPEG", quality=90, subsampling="4:2:0", optimize=False)
    assert J.encode_jpeg(rgb, 90) == buf.getvalue()


def test_oracle_vs_pillow_property():
    """Property test (hypothesis): any small RGB image, any quality -> the oracle's bytes are libjpeg-turbo's bytes."""
    PIL = pytest.importorskip("PIL.Image")
    hyp = pytest.importorskip("hypothesis")
    import io
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None, derandomize=True, database=None)
    @given(h=st.integers(1, 40), w=st.integers(1, 40), q=st.integers(1, 100), seed=st.integers(0, 2**31 - 1),
           kind=st.sampled_from(["noise", "flat", "ramp", "checker"]))
    def check(h, w, q, seed, kind):
        rng = np.random.default_rng(seed)
        if kind == "noise":
            rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        elif kind == "flat":
            rgb = np.broadcast_to(rng.integers(0, 256, 3, dtype=np.uint8), (h, w, 3)).copy()
        elif kind == "ramp":
            rgb = ((np.arange(w)[None, :, None] * 7 + np.arange(h)[:, None, None] * 5 + np.arange(3)[None, None, :] * 40) % 256).astype(np.uint8)
        else:
            rgb = (((np.arange(w)[None, :, None] + np.arange(h)[:, None, None]) % 2) * 255 * np.ones(3, dtype=np.int64)).astype(np.uint8)
        buf = io.BytesIO()
        PIL.fromarray(rgb).save(buf, "JPEG", quality=q, subsampling="4:2:0", optimize=False)
        assert J.encode_jpeg(rgb, q) == buf.getvalue()

    check()
