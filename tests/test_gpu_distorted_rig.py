"""GPU, through the C ABI, on the SECOND reference-pinned rig (tests/golden/geometry_distorted.npz: radial distortion,
off-axis projection point, skew, unequal gammas, off-centre principal point, distinct inner / outer mask centres):
  * the unwrap kernels fed with the reference's OWN LUT rows (float32 maps straight from the fixture) equal the oracle's
    fixed-point remap bit for bit, as do both views of the mirror rig (whose LUT is bit-identical to the reference's);
  * pano pixel -> angles -> bearings, midpoint triangulation and the range filter equal the fixture (rel-tol 1e-12 / 1e-9)
    and the oracle bit for bit on this panorama geometry (1200 x 133);
  * the whole hot path (sosvo_frame_pair_batch) on frames rendered through this rig equals the reference's control flow on
    the oracle (tests/refflow.py): records bit-identical."""
import numpy as np
import pytest
import torch

import oracle
from distorted_rig import D, distorted_rig

pytestmark = pytest.mark.gpu
RT = dict(rtol=1e-12, atol=1e-12)


def _to(dev, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


def _textured(rng, shape):
    base = rng.integers(0, 256, (shape[0] // 8 + 2, shape[1] // 8 + 2, shape[2]), dtype=np.uint8)
    img = np.kron(base, np.ones((8, 8, 1), dtype=np.uint8))[: shape[0], : shape[1]]
    return np.clip(img.astype(np.int16) + rng.integers(-12, 13, shape), 0, 255).astype(np.uint8)


def test_unwrap_on_the_reference_lut_rows_and_on_the_mirror_rig(ctx):
    rng = np.random.default_rng(5)
    omni = np.stack([_textured(rng, (480, 640, 3)) for _ in range(2)])
    gs = distorted_rig()
    gs.make_annulus_masks((480, 640))
    masks = np.stack([gs.top_model.mask, gs.bot_model.mask])
    # (i) the fixture's rows as the map: no model code of this repository between the reference's numbers and the kernel
    n = min(D["top_lut_rows"].shape[0], D["bot_lut_rows"].shape[0])
    mx = np.stack([D["top_lut_x"][:n], D["bot_lut_x"][:n]]).astype(np.float32)
    my = np.stack([D["top_lut_y"][:n], D["bot_lut_y"][:n]]).astype(np.float32)
    assert np.isnan(mx[1]).any() and not np.isnan(mx[0]).all()
    t_omni, t_masks, t_mx, t_my = _to(ctx.device, omni, masks, mx, my)
    pano = ctx.unwrap(t_omni, t_masks, t_mx, t_my)
    table = ctx.unwrap_prepare(t_masks, t_mx, t_my, (480, 640))
    pano_t = ctx.unwrap_table(t_omni, table)
    ctx.synchronize()
    pano, pano_t = pano.cpu().numpy(), pano_t.cpu().numpy()
    for v in range(2):
        for f in range(2):
            want = oracle.unwrap(omni[f], masks[v], mx[v], my[v])
            assert np.array_equal(pano[v, f], want) and np.array_equal(pano_t[v, f], want), (v, f)
    assert pano[0].any() and pano[1].any()
    # (ii) the mirror rig's full maps (bit-identical to the reference's LUT: tests/test_host_mirror.py)
    maps = [m.panorama.float32_maps() for m in (gs.top_model, gs.bot_model)]
    fx, fy = np.stack([maps[0][0], maps[1][0]]), np.stack([maps[0][1], maps[1][1]])
    sel = D["top_lut_rows"]
    assert np.array_equal(fx[0][sel], D["top_lut_x"].astype(np.float32), equal_nan=True)
    t_fx, t_fy = _to(ctx.device, fx, fy)
    full = ctx.unwrap(t_omni, t_masks, t_fx, t_fy)
    ctx.synchronize()
    full = full.cpu().numpy()
    assert full.shape == (2, 2, int(D["top_pano"][0]), int(D["top_pano"][1]), 3)
    for v in range(2):
        assert np.array_equal(full[v, 1], oracle.unwrap(omni[1], masks[v], fx[v], fy[v])), v
    nan_rows = D["bot_lut_nan_rows"]
    assert nan_rows.size and not full[1][:, nan_rows].any()      # rows outside the bottom mirror's own elevation range


def test_geometry_kernels_on_the_distorted_rig_fixture(ctx):
    for name in ("top", "bot"):
        rows, cols, px, hmax, _, _ = D[name + "_pano"]
        m = D["m_" + name]
        (uv,) = _to(ctx.device, m[:, :2])
        az, el, b = ctx.pano_to_bearing(uv, cols, rows, px, hmax)
        ctx.synchronize()
        az, el, b = az.cpu().numpy(), el.cpu().numpy(), b.cpu().numpy()
        assert np.array_equal(np.isnan(az), np.isnan(D["az_" + name])) and np.array_equal(np.isnan(el), np.isnan(D["el_" + name]))
        assert np.allclose(az, D["az_" + name], equal_nan=True, **RT) and np.allclose(el, D["el_" + name], equal_nan=True, **RT)
        assert np.allclose(b, D["bearing_" + name][:, :3], equal_nan=True, **RT)
        oaz, oel = oracle.pano_to_angles(m[:, 0], m[:, 1], cols, rows, px, hmax)
        assert np.array_equal(az, oaz, equal_nan=True) and np.array_equal(el, oel, equal_nan=True)
    a1, e1, a2, e2 = _to(ctx.device, D["az_top"], D["el_top"], D["az_bot"], D["el_bot"])
    X = ctx.triangulate_midpoint(a1, e1, a2, e2, D["top_F"], D["bot_F"])
    ok = ctx.range_filter(X, 500.0, 7000.0)
    ctx.synchronize()
    Xn, want = X.cpu().numpy(), D["tri_X_homo"][:, :3]
    assert np.array_equal(np.isnan(Xn).any(1), np.isnan(want).any(1))
    assert np.allclose(Xn, want, equal_nan=True, rtol=1e-9, atol=1e-7)
    assert np.array_equal(Xn, oracle.triangulate_midpoint(D["az_top"], D["el_top"], D["az_bot"], D["el_bot"], D["top_F"], D["bot_F"]),
                          equal_nan=True)
    assert np.array_equal(ok.cpu().numpy().astype(bool), D["range_ok_500_7000"])


def test_whole_path_on_frames_rendered_through_the_distorted_rig(ctx):
    import refflow
    from vo_single_camera_sos_amd import orb_pattern, synthetic
    from vo_single_camera_sos_amd.frontend import DeviceImageModel
    from vo_single_camera_sos_amd.pipeline import FramePairBatch, RigConfig
    gs = distorted_rig()
    gs.make_annulus_masks((480, 640))
    pano = gs.top_model.panorama
    geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
    rig_kw = dict(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0], min_range=500.0,
                  max_range=7000.0, stereo_min_disp=1.0, stereo_max_hdiff=2.5, f2f_max_hdiff=0.125 * 0.5 * pano.cols,
                  pct_good_matches=1.0)
    B = 3
    omni, poses = synthetic.make_frame_pairs(gs, B, seed=321)
    model = DeviceImageModel(ctx, gs, (480, 640))
    batch = FramePairBatch(ctx, model, RigConfig(**rig_kw), B, num_of_features=200, kp_cap=256, frame_cap=1024, max_iter=300,
                           seed=17)
    batch.load_frames(omni)
    got = batch.step()
    ctx.synchronize()
    got = got.cpu().numpy()
    ca, sa = orb_pattern.angle_cos_sin(-1.0)
    im_kw = dict(map_x=model.map_x.cpu().numpy(), map_y=model.map_y.cpu().numpy(), omni_masks=model.omni_masks.cpu().numpy(),
                 mask_bits=model.mask_bits_host, nmask=model.nmask, max_corners=200, pattern=model.pattern_host, cos_a=ca,
                 sin_a=sa, method="GFT", kp_cap=256)
    want = refflow.pairs_records_worker((rig_kw, im_kw, omni, batch.cfg.ransac_threshold, 300, 17))
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), np.argwhere(got != want)[:8]
    assert int((got[:, 14] == 0).sum()) >= 2 and got[:, 13].min() > 50
    for i in range(B):   # the planted motion is recovered (the renderer projects through the SAME distorted forward model)
        if got[i, 14] != 0:
            continue
        T = got[i, :12].reshape(3, 4)
        dR = T[:, :3].T @ poses[i][0]
        ang = np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))
        assert ang < 3.0 and np.linalg.norm(T[:, 3] - poses[i][1]) < 150.0, (i, ang)
