"""GPU: the C ABI used from plain C (examples/c_abi_batch.c, built with gcc against include/sosvo.h and libsosvo.so;
buffers from hipMalloc, no Python or torch in the process): sosvo_unwrap_prepare + sosvo_frame_pair_batch return the
records the ctypes path returns for the same inputs, bit for bit."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from vo_single_camera_sos_amd import synthetic
from vo_single_camera_sos_amd.frontend import DeviceImageModel
from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
from vo_single_camera_sos_amd.pipeline import FramePairBatch, RigConfig

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_matches_ctypes_path(ctx, tmp_path):
    exe = os.path.join(ROOT, "examples", "c_abi_batch")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build_example()
    B = 2
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=1200)
    gs.make_annulus_masks((480, 640))
    pano = gs.top_model.panorama
    geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
    rig = RigConfig(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0],
                    min_range=500.0, max_range=7000.0)
    model = DeviceImageModel(ctx, gs, (480, 640))
    omni, _ = synthetic.make_frame_pairs(gs, B, seed=606)
    batch = FramePairBatch(ctx, model, rig, B, num_of_features=200, kp_cap=256, frame_cap=1024, max_iter=150, seed=8)
    batch.load_frames(omni)
    want = batch.step().cpu().numpy()
    ctx.synchronize()
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(ctypes.string_at(ctypes.addressof(batch.cfg), ctypes.sizeof(batch.cfg)))
        f.write(ctypes.string_at(ctypes.addressof(batch.rig), ctypes.sizeof(batch.rig)))
        for a in (omni, model.omni_masks.cpu().numpy(), model.map_x.cpu().numpy(), model.map_y.cpu().numpy(),
                  model.mask_bits_host, model.pattern_host):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([exe, inp, outp], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(outp, dtype=np.float64).reshape(B, 16)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert "inliers" in r.stdout and (got[:, 14] == 0).all() and (got[:, 12] > 50).all()
    # "graph": the call captured into a HIP graph with the plain runtime API and replayed 20 times (exit code 7 = a replay
    # differed from the eager records)
    r3 = subprocess.run([exe, inp, outp, "graph"], capture_output=True, text=True, timeout=120)
    print(r3.stdout[-300:])
    assert r3.returncode == 0, r3.stdout[-500:] + r3.stderr[-500:]
    assert "0 of 20 replays differ" in r3.stdout
    # the same program with the batch split over two internal streams of the library
    r2 = subprocess.run([exe, inp, outp, "2"], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0, r2.stderr
    assert np.array_equal(np.fromfile(outp, dtype=np.float64).reshape(B, 16).view(np.uint64), want.view(np.uint64))
    # back-to-back enqueue + one join, and the same frames through the SEQUENCE entry points (front ends in windows of three
    # frames into a frame store, then the slot pairs): the pair batch's records, bit for bit
    for mode, text in (("enqueue", "the two record buffers agree"), ("sequence", "sequence: 4 frames")):
        r4 = subprocess.run([exe, inp, outp, mode], capture_output=True, text=True, timeout=120)
        assert r4.returncode == 0, r4.stdout[-500:] + r4.stderr[-500:]
        assert text in r4.stdout, r4.stdout
        assert np.array_equal(np.fromfile(outp, dtype=np.float64).reshape(B, 16).view(np.uint64), want.view(np.uint64)), mode
