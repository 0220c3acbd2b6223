"""GPU: sosvo_frame_pair_batch (ONE C-ABI call for the whole hot path over B pairs, include/sosvo.h) returns
exactly what the stage-by-stage sequence (ImageFrontEnd + FramePairPipeline, itself parity-tested against the
oracle in test_gpu_endtoend.py) returns on the same frames: records bit-identical, including a black frame;
argument errors are reported, not executed."""
import ctypes

import numpy as np
import pytest
import torch

from vo_single_camera_sos_amd import _lib, synthetic
from vo_single_camera_sos_amd.frontend import DeviceImageModel, ImageFrontEnd
from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
from vo_single_camera_sos_amd.pipeline import FramePairBatch, FramePairPipeline, RigConfig

pytestmark = pytest.mark.gpu


def _setup(ctx, B, width=1200):
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=width)
    gs.make_annulus_masks((480, 640))
    pano = gs.top_model.panorama
    geo = (pano.cols, pano.rows, pano.pixel_size, pano.cyl_height_max)
    rig = RigConfig(pano_top=geo, pano_bot=geo, F_top=gs.top_model.F[:3, 0], F_bot=gs.bot_model.F[:3, 0],
                    min_range=500.0, max_range=7000.0)
    model = DeviceImageModel(ctx, gs, (480, 640))
    omni, _ = synthetic.make_frame_pairs(gs, B, seed=4242)
    return model, rig, omni


def test_one_call_equals_stage_sequence(ctx):
    B, nfeat, cap = 4, 300, 320
    model, rig, omni = _setup(ctx, B)
    omni[3] = 0  # pair 1: black current frame
    fe = ImageFrontEnd(ctx, model, 2 * B, num_of_features=nfeat, kp_cap=cap)
    pipe = FramePairPipeline(ctx, rig, B, frame_cap=1024, max_iter=300, seed=11, front_end=fe)
    fe.load_frames(omni)
    pipe.step()
    want = pipe.results().cpu().numpy()
    batch = FramePairBatch(ctx, model, rig, B, num_of_features=nfeat, kp_cap=cap, frame_cap=1024, max_iter=300, seed=11)
    batch.workspace.fill_(0xA5)  # contents on entry are irrelevant
    batch.load_frames(omni)
    got = batch.step()
    ctx.synchronize()
    got = got.cpu().numpy()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert got[1, 14] == 1 and got[1, 13] == 0 and (got[[0, 2, 3], 14] == 0).all()
    # a second step on the dirty workspace gives the same records (no state carried between steps)
    got2 = batch.step()
    ctx.synchronize()
    assert np.array_equal(got2.cpu().numpy().view(np.uint64), want.view(np.uint64))
    # the same call split over the library's internal HIP streams (sosvo_frame_pair_batch_streams): same records, step
    # after step, also with more streams than the 4 pairs divide evenly into
    for ns in (2, 3):
        multi = FramePairBatch(ctx, model, rig, B, num_of_features=nfeat, kp_cap=cap, frame_cap=1024, max_iter=300, seed=11,
                               n_streams=ns)
        multi.load_frames(omni)
        for _ in range(3):
            got3 = multi.step().clone()
            ctx.synchronize()
            assert np.array_equal(got3.cpu().numpy().view(np.uint64), want.view(np.uint64)), ns


def test_streams_variant_repeats_itself_over_many_overlapped_steps(ctx):
    """150 steps of the two-stream batch call: every step's records equal the one-stream call's.  (An intermittent wrong
    match key -- a hazard in a branchy MFMA sequence of the matcher, visible only while another part's kernels shared the
    CUs -- once showed up in ~8 % of such steps and in none of the one-stream ones.)"""
    B, nfeat, cap = 4, 300, 320
    model, rig, omni = _setup(ctx, B)
    one = FramePairBatch(ctx, model, rig, B, num_of_features=nfeat, kp_cap=cap, frame_cap=1024, max_iter=300, seed=11)
    one.load_frames(omni)
    want = one.step().clone()
    ctx.synchronize()
    multi = FramePairBatch(ctx, model, rig, B, num_of_features=nfeat, kp_cap=cap, frame_cap=1024, max_iter=300, seed=11, n_streams=2)
    multi.load_frames(omni)
    bad = []
    for it in range(150):
        got = multi.step()
        ctx.synchronize()
        if not torch.equal(got, want):
            bad.append(it)
    assert not bad, bad


def test_enqueue_without_join_overlaps_calls_and_keeps_the_records(ctx):
    """sosvo_frame_pair_batch_streams_enqueue x 8 (two alternating record buffers, ONE join at the end) == the joined
    call; the join alone makes the context's stream wait (nothing is read before it)."""
    B, nfeat, cap = 4, 300, 320
    model, rig, omni = _setup(ctx, B)
    one = FramePairBatch(ctx, model, rig, B, num_of_features=nfeat, kp_cap=cap, frame_cap=1024, max_iter=300, seed=11)
    one.load_frames(omni)
    want = one.step().clone()
    ctx.synchronize()
    multi = FramePairBatch(ctx, model, rig, B, num_of_features=nfeat, kp_cap=cap, frame_cap=1024, max_iter=300, seed=11, n_streams=3)
    multi.load_frames(omni)
    bufs = [torch.zeros_like(multi.out), torch.zeros_like(multi.out)]
    for k in range(8):
        multi.enqueue(bufs[k & 1])
    multi.join()
    ctx.synchronize()
    assert torch.equal(bufs[0], want) and torch.equal(bufs[1], want)
    bufs[0].zero_()
    multi.enqueue(bufs[0])
    got = multi.step()          # a joined call right behind an un-joined one: same stream order, both complete after it
    ctx.synchronize()
    assert torch.equal(got, want) and torch.equal(bufs[0], want)


def test_multi_stream_c_call_is_capturable_after_eager_and_enqueued_calls():
    """sosvo_frame_pair_batch_streams (n_streams > 1) captured into a HIP graph AFTER eager calls -- joined and
    un-joined + join -- must not wait on a median event recorded outside the capture: the internal streams join the
    capture through sub_begin, the token chain restarts, replays are bit-identical on poisoned outputs.  Capturing while
    un-joined work is pending is refused (SOSVO_ERR_ARG), not silently mis-ordered; a different split behind an
    un-joined call joins internally and still returns the right records."""
    from vo_single_camera_sos_amd.device import Context, SosvoError
    st = torch.cuda.Stream(torch.device("cuda", 0))
    with torch.cuda.stream(st):
        c = Context(0, st)
        B, nfeat, cap = 5, 300, 320
        model, rig, omni = _setup(c, B)
        kw = dict(num_of_features=nfeat, kp_cap=cap, frame_cap=1024, max_iter=300, seed=11)
        one = FramePairBatch(c, model, rig, B, **kw)
        one.load_frames(omni)
        want = one.step().clone()
        multi = FramePairBatch(c, model, rig, B, n_streams=3, **kw)
        multi.load_frames(omni)
        other = torch.zeros_like(multi.out)
        assert torch.equal(multi.step(), want)            # eager, joined: scratch reaches its final size
        multi.enqueue(other)                              # eager, un-joined ...
        st.synchronize()
        # ... a capture now would depend on events from outside it: refused
        g_bad = torch.cuda.CUDAGraph()
        with pytest.raises(SosvoError):
            with torch.cuda.graph(g_bad, stream=st):
                multi.step()
        multi.join()                                      # ... joined: self-contained again
        st.synchronize()
        assert torch.equal(other, want)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            multi.step()
        st.synchronize()
        for _ in range(4):
            multi.out.fill_(float("nan"))
            multi.workspace.fill_(0xFF)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(multi.out, want), torch.nonzero(multi.out != want)[:8]
        # a different split right behind an un-joined call: internal join, right records, same workspace
        multi.enqueue(other)
        two = FramePairBatch(c, model, rig, B, n_streams=2, **kw)
        two.workspace = multi.workspace[: two.workspace.numel()]
        two.load_frames(omni)
        got = two.step()
        st.synchronize()
        assert torch.equal(got, want) and torch.equal(other, want)
        c.close()


def test_argument_errors(ctx):
    model, rig, omni = _setup(ctx, 1)
    batch = FramePairBatch(ctx, model, rig, 1, num_of_features=100, frame_cap=512, max_iter=50)
    small = torch.empty((batch.workspace.numel() // 2,), dtype=torch.uint8, device=ctx.device)
    with pytest.raises(_lib.SosvoError):
        ctx.frame_pair_batch(batch.rig, batch.cfg, batch.omni, model.unwrap_table, model.mask_bits, model.pattern, small)
    bad = _lib.BatchCfg()
    ctypes.memmove(ctypes.addressof(bad), ctypes.addressof(batch.cfg), ctypes.sizeof(bad))
    bad.frame_cap = 1 << 20
    assert ctx.frame_pair_batch_workspace(bad) > 0
    with pytest.raises(_lib.SosvoError):
        ctx.frame_pair_batch(batch.rig, bad, batch.omni, model.unwrap_table, model.mask_bits, model.pattern,
                             batch.workspace)


@pytest.mark.parametrize("n_streams", [1, 3])
def test_overlapped_streams_equal_single_pipeline(ctx, n_streams):
    """OverlappedFramePairs (the batch split over HIP streams, medians taking turns) returns the records of one
    FramePairPipeline over all pairs, bit for bit, step after step."""
    from vo_single_camera_sos_amd.pipeline import OverlappedFramePairs
    B, nfeat = 5, 200
    model, rig, omni = _setup(ctx, B)
    fe = ImageFrontEnd(ctx, model, 2 * B, num_of_features=nfeat, kp_cap=256)
    pipe = FramePairPipeline(ctx, rig, B, frame_cap=1024, max_iter=200, seed=3, front_end=fe)
    fe.load_frames(omni)
    pipe.step()
    want = pipe.results().cpu().numpy()
    ctx.synchronize()
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=1200)
    gs.make_annulus_masks((480, 640))
    eng = OverlappedFramePairs(0, gs, (480, 640), rig, B, n_streams=n_streams, num_of_features=nfeat, kp_cap=256,
                               frame_cap=1024, max_iter=200, seed=3)
    assert eng.S == n_streams and [p.hi - p.lo for p in eng.parts] == ([5] if n_streams == 1 else [2, 2, 1])
    eng.load_frames(omni)
    for _ in range(3):
        eng.step()
        got = eng.results().clone()
        eng.consumed()
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint64), want.view(np.uint64))
    # the PCIe-inclusive form: frames handed over in pinned host memory, copied per step into alternating device
    # buffers; frames that CHANGE from step to step give that step's records (pairs reversed here: a different result)
    pinned = torch.from_numpy(omni).pin_memory()
    swapped = torch.from_numpy(np.ascontiguousarray(omni.reshape(B, 2, *omni.shape[1:])[:, ::-1].reshape(omni.shape))).pin_memory()
    recs = []
    for k in range(4):
        eng.step_from_host(pinned if k % 2 == 0 else swapped)
        recs.append(eng.results().clone())
        eng.consumed()
    torch.cuda.synchronize()
    assert np.array_equal(recs[0].cpu().numpy().view(np.uint64), want.view(np.uint64))
    assert torch.equal(recs[2], recs[0]) and torch.equal(recs[3], recs[1]) and not torch.equal(recs[1], recs[0])
    eng.close()
