"""GPU: (1) no stage reads memory it has not written in the same step -- every output / workspace buffer of the
engine and the library's internal scratch are overwritten with 0xFF between two steps and the records must not
change; (2) the whole multi-stream step captured into a torch.cuda.CUDAGraph replays bit-identically
(OverlappedFramePairs.capture_graph: the parts' streams join the capture, no wait on an event from outside it)."""
import numpy as np
import pytest
import torch

from vo_single_camera_sos_amd import synthetic
from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
from vo_single_camera_sos_amd.pipeline import OverlappedFramePairs, RigConfig

pytestmark = pytest.mark.gpu


def _engine(n_pairs=6, n_streams=2):
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=1200)
    gs.make_annulus_masks((480, 640))
    pano = gs.top_model.panorama
    geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
    rig = RigConfig(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0], min_range=500.0,
                    max_range=7000.0)
    omni, _ = synthetic.make_frame_pairs(gs, n_pairs, seed=77)
    eng = OverlappedFramePairs(0, gs, (480, 640), rig, n_pairs, n_streams=n_streams, num_of_features=150, kp_cap=256,
                               frame_cap=1024, max_iter=300, seed=5)
    eng.load_frames(omni)
    return eng


def _poison(eng):
    """0xFF into everything a step writes (never into the inputs: omni frames, model constants, index tables)."""
    def fill(t):
        t.view(torch.uint8).fill_(0xFF)
    for p in eng.parts:
        with torch.cuda.stream(p.stream):
            fe, pipe = p.fe, p.pipe
            for t in (fe.gray, fe.kp, fe.n, fe.status, fe.desc, pipe.s_keys, pipe.s_order, pipe._d2, pipe._M2, pipe._k2,
                      pipe._o2, pipe.T, pipe.lm_cost, pipe.lm_iters):
                fill(t)
            for d in (pipe.frames, pipe.corr, pipe.ransac):
                for k, t in d.items():
                    if k not in ("d_top", "d_bot", "M"):      # views of _d2 / _M2
                        fill(t)
            p.ctx.debug_fill_scratch(0xFF)
    with torch.cuda.stream(torch.cuda.current_stream(eng.device)):
        eng.out.fill_(float("nan"))
    torch.cuda.synchronize(eng.device)


def test_no_stage_reads_memory_it_did_not_write():
    eng = _engine()
    eng.step()
    rec0 = eng.results().clone()
    eng.consumed()
    torch.cuda.synchronize()
    assert int((rec0[:, 14] == 0).sum()) >= 5, rec0[:, 12:]
    for _ in range(2):
        _poison(eng)
        eng.step()
        rec1 = eng.results().clone()
        eng.consumed()
        torch.cuda.synchronize()
        assert torch.equal(rec0, rec1), torch.nonzero(rec0 != rec1)[:8]
    eng.close()


def test_cuda_graph_replay_is_bit_identical():
    eng = _engine()
    for _ in range(2):                       # eager: scratch reaches its final size, records to compare with
        eng.step()
        rec0 = eng.results().clone()
        eng.consumed()
    torch.cuda.synchronize()
    g = eng.capture_graph()
    for _ in range(4):
        _poison(eng)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(eng.out, rec0), torch.nonzero(eng.out != rec0)[:8]
    eng.step()                               # eager steps still work after a capture
    rec2 = eng.results().clone()
    eng.consumed()
    torch.cuda.synchronize()
    assert torch.equal(rec2, rec0)
    eng.close()
