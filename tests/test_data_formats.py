"""Data formats either side of the hot path against bytes / values produced by the reference."""
import numpy as np
import pytest

from ocflow_amd.data import flow_io


def test_flo_wire_format(golden, tmp_path):
    g = golden("epe")
    raw = g["flo_bytes"].tobytes()
    assert flow_io.encode_flo(g["flo"]) == raw  # byte-identical to the reference's save_flow
    np.testing.assert_array_equal(flow_io.decode_flo(raw), g["flo_read"])  # and its read_flow
    p = tmp_path / "a.flo"
    flow_io.save_flow(str(p), g["flo"][..., 0], g["flo"][..., 1])
    np.testing.assert_array_equal(flow_io.read_flow(str(p)), g["flo"])
    with pytest.raises(ValueError):
        flow_io.decode_flo(b"\x00" * 16)


def test_epe(golden):
    g = golden("epe")
    assert abs(flow_io.flow_error(g["tu"], g["tv"], g["u"], g["v"]) - float(g["epe"])) < 1e-6
    assert abs(flow_io.flow_error(g["tu"], g["tv"], g["u"], g["v"], g["occ"]) - float(g["epe_occ"])) < 1e-6


def test_shape_rules():
    assert flow_io.render_size(436, 1024) == (384, 1024)   # Sintel
    assert flow_io.render_size(375, 1242) == (320, 1216)   # KITTI (BASELINE configs[3])
    img = np.array([[[0, 128, 255]]], dtype=np.uint8)
    out = flow_io.normalize_image(img)
    assert out.shape == (3, 1, 1) and out[0, 0, 0] == -1.0 and out[2, 0, 0] == 1.0
