"""GPU: SEQUENCE MODE of the SOS hot path (include/sosvo.h "Sequence mode", pipeline.SequenceEngine, run_VO's
frame_window): the reference's VO loop computes every frame's front end once and tracks it against the current keyframe
(omnistereo/pose_est_tools.py:1416-1628).  Checked here:
  * the store-based entry points give the records of the pair batch (sosvo_frame_pair_batch, itself bit-exact against
    the CPU oracle in test_gpu_batch_call.py / test_gpu_endtoend.py) on the same frames, bit for bit, whatever the window;
  * run_VO on a 200-frame synthetic sequence writes the byte-identical estimated_frame_poses_TUM.txt for windows of 32, 5
    and 1 frames (1 = the per-frame use of the same entry points), and the host-array mirror path agrees to rounding."""
import os

import numpy as np
import pytest

from vo_single_camera_sos_amd import synthetic
from vo_single_camera_sos_amd.frontend import DeviceImageModel
from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
from vo_single_camera_sos_amd.pipeline import FramePairBatch, RigConfig, SequenceEngine

pytestmark = pytest.mark.gpu


def _rig():
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=1200)
    gs.make_annulus_masks((480, 640))
    pano = gs.top_model.panorama
    geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
    rig_kw = dict(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0], min_range=500.0,
                  max_range=7000.0, stereo_min_disp=1.0, stereo_max_hdiff=2.5, f2f_max_hdiff=0.125 * 0.5 * pano.cols,
                  pct_good_matches=1.0)
    return gs, rig_kw


def _sequence_frames(gs, n, seed):
    room = synthetic.Room(seed=seed)
    poses = synthetic.trajectory(n, seed, max_t=40.0, max_deg=2.0)
    rng = np.random.default_rng(seed + 1)
    return np.stack([synthetic.render_omni(gs, room, R, t, 2.0, rng) for R, t in poses]), poses


def test_sequence_engine_equals_the_pair_batch_for_every_window(ctx):
    gs, rig_kw = _rig()
    n = 7
    frames, _ = _sequence_frames(gs, n, seed=61)
    model = DeviceImageModel(ctx, gs, (480, 640))
    kw = dict(num_of_features=300, kp_cap=320, frame_cap=1024, max_iter=210, adaptive=True, ransac_solver="GP3P")
    # reference records: the pair batch on (frame t - 1, frame t) with the seed the sequence gives frame t
    want = []
    for t in range(1, n):
        pb = FramePairBatch(ctx, model, RigConfig(**rig_kw), 1, seed=t - 1, **kw)
        pb.load_frames(frames[t - 1:t + 1])
        want.append(pb.step().cpu().numpy()[0].copy())
    want = np.stack(want)
    assert (want[:, 14] == 0).all() and (want[:, 12] > 50).all(), want[:, 12:16]
    counts_ref = None
    for window in (1, 2, 3, 7, 16):
        eng = SequenceEngine(ctx, model, RigConfig(**rig_kw), window=window, **kw)
        got, counts = [], []
        for w0 in range(0, n, window):
            infos = eng.push_window(list(frames[w0:w0 + window]))
            for i, info in enumerate(infos):
                t = w0 + i
                counts.append(info["count"])
                assert info["seed"] == max(t - 1, 0)
                if t == 0:
                    assert info["spec"] is None
                    continue
                got.append(info["spec"])
        got = np.stack(got)
        assert np.array_equal(got, want), (window, np.argwhere(got != want)[:6])
        assert counts_ref is None or counts == counts_ref
        counts_ref = counts
        # the same windows staged and copied AHEAD on the engine's copy stream (what run_VO's helper thread does), through
        # alternating pinned / device input buffers, from another host thread: the same records
        if window in (2, 3):
            import threading
            eng3 = SequenceEngine(ctx, model, RigConfig(**rig_kw), window=window, **kw)
            got3, buf = [], 0
            chunks = [list(frames[w0:w0 + window]) for w0 in range(0, n, window)]

            def stage(c, b):
                eng3.stage_host(c, b)
                eng3.upload_staged(b, len(c))
            stage(chunks[0], 0)
            for k, c in enumerate(chunks):
                th = None
                if k + 1 < len(chunks):     # the next window goes up while this one is pushed
                    th = threading.Thread(target=stage, args=(chunks[k + 1], 1 - buf))
                    th.start()
                infos = eng3.push_staged(buf, len(c))
                got3 += [info["spec"] for info in infos if info["spec"] is not None]
                if th is not None:
                    th.join()
                buf = 1 - buf
            assert np.array_equal(np.stack(got3), want), window
        # windows enqueued AHEAD (run_VO's loop: window k + 1 is on the GPU while the host works through window k), with a
        # keyframe whose half is refilled under it and a serial call behind the next window's work
        if window == 2:
            eng4 = SequenceEngine(ctx, model, RigConfig(**rig_kw), window=2, **kw)
            with pytest.raises(RuntimeError):
                eng4.collect()                                    # nothing pending
            eng4.stage_host(list(frames[0:2]), 0)
            eng4.enqueue_staged(0, 2)
            with pytest.raises(RuntimeError):
                eng4.enqueue_staged(1, 2)                         # one window may be pending
            i0 = eng4.collect()
            eng4.stage_host(list(frames[2:4]), 1)
            eng4.enqueue_staged(1, 2)                             # window 1 goes ahead ...
            eng4.promote(i0[1]["slot"])                           # ... while the host makes frame 1 the keyframe
            i1 = eng4.collect()
            eng4.stage_host(list(frames[4:6]), 0)
            eng4.enqueue_staged(0, 2)                             # refills the half frame 1 lives in: its record moves first
            rec = eng4.track(eng4.key_slot, i1[1]["slot"], seed=2)   # frame 3 against keyframe 1, behind window 2's work
            pb = FramePairBatch(ctx, model, RigConfig(**rig_kw), 1, seed=2, **kw)
            pb.load_frames(frames[[1, 3]])
            assert np.array_equal(rec, pb.step().cpu().numpy()[0])
            i2 = eng4.collect()
            got4 = [i["spec"] for i in i0 + i1 + i2 if i["spec"] is not None]
            assert np.array_equal(np.stack(got4), want[:5])
        # a serial call against the keyframe slot: frame 1 promoted, frame 3 tracked against it (not its predecessor)
        if window >= 4:
            eng2 = SequenceEngine(ctx, model, RigConfig(**rig_kw), window=window, **kw)
            infos = eng2.push_window(list(frames[:4]))
            eng2.promote(infos[1]["slot"])
            rec = eng2.track(eng2.key_slot, infos[3]["slot"], seed=2)
            pb = FramePairBatch(ctx, model, RigConfig(**rig_kw), 1, seed=2, **kw)
            pb.load_frames(frames[[1, 3]])
            assert np.array_equal(rec, pb.step().cpu().numpy()[0]) and eng2.serial_calls == 1
    assert min(counts_ref) > 100


def test_sequence_entry_points_refuse_bad_arguments(ctx):
    from vo_single_camera_sos_amd.device import SosvoError
    gs, rig_kw = _rig()
    model = DeviceImageModel(ctx, gs, (480, 640))
    eng = SequenceEngine(ctx, model, RigConfig(**rig_kw), window=2, num_of_features=100, kp_cap=128, frame_cap=512)
    with pytest.raises(ValueError):
        eng.push_window([np.zeros((480, 640, 3), np.uint8)] * 3)              # more frames than the window
    with pytest.raises(SosvoError):
        eng.track(eng.slots, 0, 0)                                            # slot outside the store
    with pytest.raises(SosvoError):
        ctx.sequence_copy_slot(eng.cfg, eng.W, eng.slots, 0, eng.slots, eng.workspace)
    small = eng.workspace[:1024]
    with pytest.raises(SosvoError):
        ctx.sequence_copy_slot(eng.cfg, eng.W, eng.slots, 0, 1, small)        # workspace too small
    assert ctx.sequence_workspace(eng.cfg, 4, 3) == 0                          # fewer slots than the window


def test_run_vo_pose_file_does_not_depend_on_the_frame_window(ctx, tmp_path):
    """A 200-frame sequence through demo_vo_sos.main_sos_vo with frame windows of 32, 5 and 1: byte-identical
    estimated_frame_poses_TUM.txt and keyframe_ids.txt; the per-frame mirror path (frame_window 0: StereoPanoramicFrame on
    host arrays) agrees on the first 25 frames to 1e-6; the trajectory follows the planted one."""
    import demo_vo_sos
    from vo_single_camera_sos_amd.omnistereo import transformations as tr
    gs, _ = _rig()
    n = 200
    seq = str(tmp_path / "seq200")
    poses = synthetic.write_sos_sequence(seq, gs, n_frames=n, seed=77, max_t=40.0, max_deg=2.0)
    gums = os.path.join(seq, "gums-calibrated.json")
    outs, texts, kfs = {}, {}, {}
    for w in (32, 5, 1):
        outs[w] = demo_vo_sos.main_sos_vo([seq, "--calibrated_gums_file", gums, "--frame_window", str(w)])
        assert outs[w]["tracked"] == n - 1 and outs[w]["sequence_mode"]["frame_window"] == w
        texts[w] = open(os.path.join(seq, "results-omni", "estimated_frame_poses_TUM.txt")).read()
        kfs[w] = open(os.path.join(seq, "results-omni", "keyframe_ids.txt")).read()
    assert texts[32] == texts[5] == texts[1] and len(texts[32].splitlines()) == n
    assert kfs[32] == kfs[5] == kfs[1]
    assert outs[32]["sequence_mode"]["windows"] == -(-n // 32)
    # the speculation covers every frame whose reference (the newest keyframe before it) is its predecessor or -- windows of
    # two frames and more -- the frame before that; the others took one serial call each
    kf = sorted(outs[32]["keyframe_ids"])
    ref = lambda i: max(k for k in kf if k < i)  # noqa: E731
    assert outs[1]["sequence_mode"]["serial_tracking_calls"] == sum(1 for i in range(1, n) if ref(i) != i - 1) > 5
    for w in (32, 5):
        assert outs[w]["sequence_mode"]["serial_tracking_calls"] == sum(1 for i in range(1, n) if ref(i) < i - 2)
    # mirror path on a prefix
    m = demo_vo_sos.main_sos_vo([seq, "--calibrated_gums_file", gums, "--frame_window", "0", "--last_image_index", "25"])
    assert "sequence_mode" not in m and len(m["poses"]) == 25
    for (i, Ta), (j, Tb) in zip(m["poses"], outs[32]["poses"][:25]):
        assert i == j and np.allclose(Ta, Tb, rtol=1e-6, atol=1e-9), (i, np.abs(Ta - Tb).max())
    # follows the planted trajectory (5-degree RANSAC threshold: degrees / centimetres per keyframe hop, drift accumulates)
    T_gt = np.identity(4)
    T_gt[:3, :3], T_gt[:3, 3] = poses[n - 1][0], poses[n - 1][1] * 1e-3
    E = tr.rpe(T_gt, outs[32]["poses"][n - 1][1])
    assert tr.rpe_rotation_metric(E) < np.deg2rad(25.0) and tr.rpe_translation_metric(E) < 2.0, E


def test_run_vo_stops_at_an_untrackable_frame_like_the_mirror_path(ctx, tmp_path):
    """A black frame in the middle of a sequence has no correspondences: the reference's loop warns and stops
    (pose_est_tools.py:1490-1494).  Sequence mode (any window) and the per-frame mirror path stop at the same frame with the
    same poses before it."""
    import demo_vo_sos
    from vo_single_camera_sos_amd.omnistereo.common_cv import imwrite
    gs, _ = _rig()
    n = 9
    seq = str(tmp_path / "seq_black")
    synthetic.write_sos_sequence(seq, gs, n_frames=n, seed=5, max_t=40.0, max_deg=2.0)
    imwrite(os.path.join(seq, "omni", "image-0005.png"), np.zeros((480, 640, 3), np.uint8))
    gums = os.path.join(seq, "gums-calibrated.json")
    outs = {}
    for w in (4, 1, 0):
        with pytest.warns(UserWarning):
            outs[w] = demo_vo_sos.main_sos_vo([seq, "--calibrated_gums_file", gums, "--frame_window", str(w),
                                               "--use_multithreads_for_VO", "false"])
    for w in (4, 1, 0):
        assert [p[0] for p in outs[w]["poses"]] == [0, 1, 2, 3, 4], (w, [p[0] for p in outs[w]["poses"]])
    for (i, Ta), (_, Tb), (_, Tc) in zip(outs[4]["poses"], outs[1]["poses"], outs[0]["poses"]):
        assert np.array_equal(Ta, Tb) and np.allclose(Ta, Tc, rtol=1e-6, atol=1e-9), i


def test_rgbd_sequence_engine_and_demo(ctx, tmp_path):
    """RGB-D sequence mode: (i) RGBDSequenceEngine's records equal sosvo_rgbd_pair_batch on (frame t - 1, frame t) with the
    sequence's seed, for several windows, bit for bit; (ii) demo_vo_rgbd.main_rgbd_vo writes the byte-identical pose file for
    frame windows 8 and 1, and the per-frame mirror path (frame_window 0) gives the same poses (its frames are computed by the
    same kernels, one frame per call)."""
    import demo_vo_rgbd
    from vo_single_camera_sos_amd.pipeline import RGBDCamConfig, RGBDPairBatch, RGBDSequenceEngine
    n = 12
    seq = str(tmp_path / "seq_rgbd")
    synthetic.write_rgbd_sequence(seq, n_frames=n, seed=41, max_t=40.0, max_deg=2.0, depth_is_Z=False)
    from vo_single_camera_sos_amd.omnistereo.common_cv import get_depthmap_float32_from_png, get_images, imread
    names = get_images(os.path.join(seq, "rgbd", "rgb", "*.png"), indices_list=None, return_names_only=True)
    dnames = get_images(os.path.join(seq, "rgbd", "depth", "*.png"), indices_list=None, return_names_only=True)
    bgr = np.stack([np.ascontiguousarray(imread(f)[..., ::-1]) for f in names])       # as run_VO hands them over (:1430)
    depth = np.stack([get_depthmap_float32_from_png(f, 1.0 / 1000.0) for f in dnames])
    cam = RGBDCamConfig(fx=554.256258, fy=554.256258, center_x=319.5, center_y=239.5, depth_is_Z=False, min_range=0.8, max_range=7.0)
    kw = dict(num_of_features=1000, max_iter=210, adaptive=True, pose_est_algorithm="EPNP")
    want = []
    for t in range(1, 6):
        pb = RGBDPairBatch(ctx, cam, 1, seed=t - 1, **kw)
        pb.load_frames(bgr[t - 1:t + 1], depth[t - 1:t + 1])
        want.append(pb.step().cpu().numpy()[0].copy())
    want = np.stack(want)
    assert (want[:, 14] == 0).all() and (want[:, 12] > 100).all(), want[:, 12:16]
    for window in (1, 4, 6):
        eng = RGBDSequenceEngine(ctx, cam, window=window, **kw)
        got = []
        for w0 in range(0, 6, window):
            for info in eng.push_window([(bgr[i], depth[i]) for i in range(w0, min(6, w0 + window))]):
                if info["spec"] is not None:
                    got.append(info["spec"])
        assert np.array_equal(np.stack(got), want), window
    outs, texts = {}, {}
    for w in (8, 1, 0):
        outs[w] = demo_vo_rgbd.main_rgbd_vo([seq, "--is_synthetic", "true", "--frame_window", str(w)])
        assert outs[w]["tracked"] == n - 1, (w, outs[w]["tracked"])
        texts[w] = open(os.path.join(seq, "results-rgbd", "estimated_frame_poses_TUM.txt")).read()
    assert texts[8] == texts[1] and "sequence_mode" in outs[8] and "sequence_mode" not in outs[0]
    for (i, Ta), (_, Tb) in zip(outs[8]["poses"], outs[0]["poses"]):
        assert np.allclose(Ta, Tb, rtol=1e-9, atol=1e-12), (i, np.abs(Ta - Tb).max())
