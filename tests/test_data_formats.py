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


# ------------------------------------------------------------------------------------------------ datasets / data module (8f-3)
def _sintel_tree(root, scenes=(("alley_1", 4), ("market_2", 3)), h=100, w=150, seed=0):
    """a synthetic MPI-Sintel layout: <dstype>/<scene>/frame_XXXX.png, flow/<scene>/frame_XXXX.flo, occlusions/<scene>/frame_XXXX.png
    (n frames -> n - 1 flows / occlusion maps per scene); returns the raw arrays keyed by file stem"""
    from PIL import Image
    rs = np.random.RandomState(seed)
    raw = {}
    for scene, n in scenes:
        for sub in ("clean", "final", "flow", "occlusions"):
            (root / sub / scene).mkdir(parents=True)
        for i in range(1, n + 1):
            for ds in ("clean", "final"):
                img = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
                Image.fromarray(img).save(root / ds / scene / f"frame_{i:04d}.png")
                raw[(ds, scene, i)] = img
            if i < n:
                fl = (rs.randn(h, w, 2) * 3).astype(np.float32)
                flow_io.save_flow(str(root / "flow" / scene / f"frame_{i:04d}.flo"), fl)
                oc = (rs.rand(h, w) < 0.3).astype(np.uint8) * 255
                Image.fromarray(oc).save(root / "occlusions" / scene / f"frame_{i:04d}.png")
                raw[("flow", scene, i)], raw[("occ", scene, i)] = fl, oc
    return raw


def test_sintel_datasets_follow_the_reference_sample_rules(tmp_path):
    """models/data/datasets.py:113-330 restated (the reference's own classes cannot run here: imageio / cv2 / torchvision are absent, so this
    is checked against an independent numpy statement of the rules, not a golden): pairing of frame i with i + 1 per flow file in sorted
    order, centre crop to multiples of 64 (:146-150, :48-53), ToTensor + Normalize(0.5, 0.5), flow cropped and channel-first, stack / cat,
    occlusion > 0.5 -> 1 (:260-261), `replicates`."""
    import torch
    from ocflow_amd.data import datasets as D
    raw = _sintel_tree(tmp_path)
    ds = D.MpiSintelClean(root=str(tmp_path), stack_imgs=False)
    assert len(ds) == 5 and ds.render_size == [64, 128]
    order = [("alley_1", 1), ("alley_1", 2), ("alley_1", 3), ("market_2", 1), ("market_2", 2)]
    y0, x0 = (100 - 64) // 2, (150 - 128) // 2
    for idx, (scene, i) in enumerate(order):
        imgs, flow = ds[idx]
        assert imgs.shape == (6, 64, 128) and imgs.dtype == torch.float32 and flow.shape == (2, 64, 128)
        for k, fi in enumerate((i, i + 1)):
            want = (raw[("clean", scene, fi)][y0:y0 + 64, x0:x0 + 128].astype(np.float32) / 255.0 - 0.5) / 0.5
            np.testing.assert_allclose(imgs[3 * k:3 * k + 3].numpy(), want.transpose(2, 0, 1), rtol=0, atol=1e-7)
        np.testing.assert_array_equal(flow.numpy(), raw[("flow", scene, i)][y0:y0 + 64, x0:x0 + 128].transpose(2, 0, 1))
    assert D.MpiSintelFinal(root=str(tmp_path), stack_imgs=True)[0][0].shape == (2, 3, 64, 128)
    assert len(D.MpiSintelClean(root=str(tmp_path), replicates=3)) == 15
    both = D.MpiSintelCleanFlowOcc(root=str(tmp_path), stack_imgs=False)
    imgs, flow, occ = both[3]
    want_occ = (raw[("occ", "market_2", 1)][y0:y0 + 64, x0:x0 + 128].astype(np.float32) > 0.5).astype(np.float32)
    np.testing.assert_array_equal(occ.numpy()[0], want_occ)
    assert set(np.unique(occ.numpy())) <= {0.0, 1.0}
    assert D.MpiSintelFinalOcc(root=str(tmp_path), stack_imgs=False)[0][1].shape == (1, 64, 128)
    with pytest.raises(FileNotFoundError):
        D.MpiSintelClean(root=str(tmp_path / "nothing"))


def test_resize_rules_and_device_side_batch_preparation(tmp_path):
    """image_size: bilinear resize with half-pixel centres (cv2.INTER_LINEAR's definition; cv2 is absent: unpinned) and the flow rescaled
    by the size ratio (flow_utils.py:120-138); `raw=True` + prepare_batch_on_device reproduces the host sample path as batched tensor
    operations (exactly without a resize; within one uint8 rounding step with one)"""
    import torch
    from ocflow_amd.data import datasets as D
    _sintel_tree(tmp_path)
    const = np.zeros((10, 12, 2), np.float32)
    const[..., 0], const[..., 1] = 2.0, -1.0
    r = D.resize_flow(const, 20, 36)
    assert r.shape == (20, 36, 2) and np.allclose(r[..., 0], 2.0 * 36 / 12) and np.allclose(r[..., 1], -1.0 * 20 / 10)
    ramp = np.tile(np.arange(8, dtype=np.float32)[None, :, None], (4, 1, 1))
    up = D.resize_linear(ramp, 4, 16)  # half-pixel centres: output x maps to (x + 0.5) / 2 - 0.5, clamped at the border
    assert np.allclose(up[0, :4, 0], [0.0, 0.25, 0.75, 1.25]) and np.allclose(up[0, -2:, 0], [6.75, 7.0])
    for size in (None, (32, 96)):
        host = D.MpiSintelCleanFlowOcc(root=str(tmp_path), image_size=size, stack_imgs=False)
        rawds = D.MpiSintelCleanFlowOcc(root=str(tmp_path), stack_imgs=False, raw=True)
        idx = [0, 2, 4]
        frames = torch.stack([rawds[i][0] for i in idx])
        flow = torch.stack([rawds[i][1] for i in idx])
        occ = torch.stack([rawds[i][2] for i in idx])
        assert frames.dtype == torch.uint8 and frames.shape == (3, 100, 150, 6)
        imgs_d, flow_d, occ_d = D.prepare_batch_on_device(frames, flow, occ, size, device="cpu", channels_last=False)
        for k, i in enumerate(idx):
            hi, hf, ho = host[i]
            tol = 0.0 if size is None else 2.0 / 255 + 1e-6
            assert float((imgs_d[k] - hi).abs().max()) <= tol
            np.testing.assert_allclose(flow_d[k].numpy(), hf.numpy(), rtol=1e-5, atol=1e-5)
            assert float((occ_d[k] - ho).abs().mean()) <= (0.0 if size is None else 0.02)


def test_dataset_module_splits_like_the_reference(tmp_path):
    """lightning_datamodule.py:50-69: ceil(0.8 n) / ceil(0.1 n) / rest through random_split(generator seed 42); overfit = the same set thrice"""
    import torch
    from ocflow_amd.data import datasets as D
    _sintel_tree(tmp_path, scenes=(("a", 8), ("b", 7)))  # 7 + 6 = 13 samples
    dm = D.DatasetModule(root=str(tmp_path), batch_size=4, dataset_name="MpiSintelFinal", num_workers=0)
    dm.setup()
    assert [len(dm.datasets[k]) for k in ("train", "val", "test")] == [11, 2, 0]
    ref = torch.utils.data.random_split(range(13), [11, 2, 0], generator=torch.Generator().manual_seed(42))
    assert list(dm.datasets["train"].indices) == list(ref[0].indices)
    batch = next(iter(dm.val_dataloader()))
    assert batch[0].shape == (2, 6, 64, 128) and batch[1].shape == (2, 2, 64, 128)
    dm2 = D.DatasetModule(root=str(tmp_path), batch_size=2, dataset_name="MpiSintelCleanFlowOcc", num_workers=0, overfit=True, raw=True, image_size=(64, 64))
    dm2.setup()
    assert dm2.datasets["train"] is dm2.datasets["test"]
    imgs, flow, occ = dm2.to_device(next(iter(dm2.train_dataloader())), "cpu")
    assert imgs.shape == (2, 6, 64, 64) and flow.shape == (2, 2, 64, 64) and occ.shape == (2, 1, 64, 64)
    with pytest.raises(ValueError, match="Unsupported dataset type"):
        D.DatasetModule(root=str(tmp_path), dataset_name="KITTI").setup()


def test_flying_chairs_layouts(tmp_path):
    """FlyingChairs (`*.ppm` + `*.flo`) and FlyingChairs2 (`-img_0/1.png`, `-flow_01.flo`, `-occ_01.png`: the dataset of the reference's shipped
    unsupervised_config.yml) pair consecutive sorted images, crop to multiples of 64, and return (images, flow[, occ]) with the shared sample
    rules (datasets.py:465-530, 587-669); the decode-only mode + device-side preparation gives the same batch"""
    from PIL import Image
    from ocflow_amd.data import flow_io
    from ocflow_amd.data.datasets import DATASETS, FlyingChairs, FlyingChairs2, prepare_batch_on_device
    rs = np.random.RandomState(5)
    c1, c2 = tmp_path / "chairs", tmp_path / "chairs2"
    c1.mkdir(), c2.mkdir()
    h, w = 70, 130  # -> render size 64 x 128
    for i in range(3):
        for j in (1, 2):
            Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(c1 / f"{i:05d}_img{j}.ppm")
        flow_io.save_flow(str(c1 / f"{i:05d}_flow.flo"), rs.randn(h, w, 2).astype(np.float32))
        for j in (0, 1):
            Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(c2 / f"{i:07d}-img_{j}.png")
        flow_io.save_flow(str(c2 / f"{i:07d}-flow_01.flo"), rs.randn(h, w, 2).astype(np.float32))
        Image.fromarray(((rs.rand(h, w) < 0.3) * 255).astype(np.uint8)).save(c2 / f"{i:07d}-occ_01.png")
    assert DATASETS["FlyingChairs2"] is FlyingChairs2
    d1 = FlyingChairs(root=str(c1), stack_imgs=False)
    imgs, flow = d1[1]
    assert len(d1) == 3 and imgs.shape == (6, 64, 128) and flow.shape == (2, 64, 128) and float(imgs.min()) >= -1.0 and float(imgs.max()) <= 1.0
    ref = np.asarray(Image.open(c1 / "00001_img1.ppm"))[3:67, 1:129]
    np.testing.assert_allclose(imgs[:3].numpy(), (ref.transpose(2, 0, 1) / 255.0 - 0.5) / 0.5, atol=1e-6)
    d2 = FlyingChairs2(root=str(c2), stack_imgs=False)
    imgs2, flow2, occ2 = d2[2]
    assert imgs2.shape == (6, 64, 128) and flow2.shape == (2, 64, 128) and occ2.shape == (1, 64, 128)
    assert set(np.unique(occ2.numpy()).tolist()) <= {0.0, 1.0} and 0.1 < float(occ2.mean()) < 0.5
    np.testing.assert_array_equal(occ2[0].numpy(), (np.asarray(Image.open(c2 / "0000002-occ_01.png"))[3:67, 1:129] > 0).astype(np.float32))
    raw = FlyingChairs2(root=str(c2), raw=True)[2]
    x, fl, oc = prepare_batch_on_device(raw[0][None], raw[1][None], raw[2][None], None, "cpu", channels_last=False)
    np.testing.assert_allclose(x[0].numpy(), imgs2.numpy(), atol=1e-6)
    np.testing.assert_array_equal(fl[0].numpy(), flow2.numpy())
    np.testing.assert_array_equal(oc[0].numpy(), occ2.numpy())
    with pytest.raises(FileNotFoundError, match="img_0"):
        FlyingChairs2(root=str(tmp_path / "empty"))
