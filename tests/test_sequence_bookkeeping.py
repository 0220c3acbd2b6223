"""Host logic of the sequence engines (pipeline._SequenceBase) on the CPU, with the device calls replaced by a recorder:
window halves, speculative (reference, current, seed) triples, the lazy keyframe promotion, and the split of push_staged into
enqueue_staged / collect that run_VO uses to put window k + 1 on the GPU before it works through window k.  What must hold
whatever the order of the host's calls: a tracking record is asked for with the triple the serial loop would use, and a
promoted frame's record is copied to the keyframe slot before the half it lives in is refilled."""
import numpy as np
import pytest
import torch

from vo_single_camera_sos_amd.pipeline import _SequenceBase


class _Stream(object):
    def wait_event(self, ev):
        pass


class _Ctx(object):
    stream = _Stream()
    device = "cpu"
    synced = 0

    def synchronize(self):
        self.synced += 1


class Recorder(_SequenceBase):
    """Frames are integers; a slot of the store holds the frame number written there; a record is (ref frame, cur frame, seed)."""

    def __init__(self, window):
        self.ctx = _Ctx()
        self.log = []
        self.store = {}
        self.staged = {0: [], 1: []}
        self._init_windows(window)
        self._win_events = (_Ev(), _Ev())          # (no GPU events here)
        self.spec = torch.zeros((self.W, 16), dtype=torch.float64)
        self.spec2 = torch.zeros((self.W, 16), dtype=torch.float64)
        self.one = torch.zeros((1, 16), dtype=torch.float64)

    def _stage_host(self, images, buf):
        self.staged[buf] = list(images)

    def _upload(self, buf, n):
        self.log.append(("upload", buf, n))

    def _front_end(self, n, first, buf):
        self.log.append(("front_end", first, n))
        for i in range(n):
            self.store[first + i] = self.staged[buf][i]

    def _track(self, ref_slots, cur_slots, seed, out):
        for k, (r, c) in enumerate(zip(ref_slots, cur_slots)):
            out[k, 0], out[k, 1], out[k, 2] = self.store[r], self.store[c], seed + k
        self.log.append(("track", tuple(ref_slots), tuple(cur_slots), seed))

    def _counts(self, first, n):
        return [100 + self.store[first + i] for i in range(n)]

    def _copy(self, src, dst):
        self.log.append(("copy", src, dst))
        self.store[dst] = self.store[src]


class _Ev(object):
    def record(self, stream=None):
        pass

    def elapsed_time(self, other):
        return 0.0


def _windows(n, w):
    return [list(range(k, min(n, k + w))) for k in range(0, n, w)]


@pytest.mark.parametrize("window", [1, 2, 3, 5, 8])
def test_speculative_triples_are_the_serial_loops(window):
    eng = Recorder(window)
    buf, t = 0, 0
    for chunk in _windows(19, window):
        eng.stage_host(chunk, buf)
        infos = eng.push_staged(buf, len(chunk))
        for i, info in enumerate(infos):
            assert info["count"] == 100 + t and info["seed"] == max(t - 1, 0)
            if t == 0:
                assert info["spec"] is None
            else:   # frame t against its predecessor with seed t - 1
                assert tuple(info["spec"][:3]) == (t - 1, t, t - 1), (window, t, info["spec"][:3])
            if info["spec2"] is not None:   # frame t against frame t - 2, same seed
                assert tuple(info["spec2"][:3]) == (t - 2, t, t - 1), (window, t)
            elif window >= 3:               # (the frame two back is in the store unless its half is the one being refilled)
                assert t < 2, (window, t)
            t += 1
        buf = 1 - buf
    assert t == 19


@pytest.mark.parametrize("window", [2, 4])
def test_enqueue_ahead_gives_the_same_records_and_moves_the_keyframe_in_time(window):
    chunks = _windows(4 * window + 1, window)

    def run(ahead):
        eng = Recorder(window)
        records, serial = [], []
        eng.stage_host(chunks[0], 0)
        eng.enqueue_staged(0, len(chunks[0]))
        buf = 0
        for k, chunk in enumerate(chunks):
            infos = eng.collect()
            nxt = chunks[k + 1] if k + 1 < len(chunks) else None
            if ahead and nxt is not None:
                eng.stage_host(nxt, 1 - buf)
                eng.enqueue_staged(1 - buf, len(nxt))
            for info in infos:
                records.append((None if info["spec"] is None else tuple(info["spec"][:3]), info["count"], info["slot"]))
            # the host's policy: the FIRST frame of every window becomes the keyframe; the last frame of the window is then
            # tracked against it serially (a reference that is not its predecessor)
            eng.promote(infos[0]["slot"])
            if len(infos) > 1:
                rec = eng.track(eng.key_slot, infos[-1]["slot"], seed=7)
                serial.append(tuple(rec[:3]))
            if not ahead and nxt is not None:
                eng.stage_host(nxt, 1 - buf)
                eng.enqueue_staged(1 - buf, len(nxt))
            buf = 1 - buf
        return records, serial, eng.log

    rec_a, ser_a, log_a = run(True)
    rec_b, ser_b, _ = run(False)
    assert rec_a == rec_b and ser_a == ser_b
    # the serial records really are (keyframe = first frame of the window, last frame of the window)
    assert ser_a == [(c[0], c[-1], 7) for c in chunks if len(c) > 1]
    assert any(ev[0] == "copy" for ev in log_a)
    # a front end never refills a half that holds a promoted frame whose record has not been copied out
    eng = Recorder(window)   # (replay with the invariant checked at every front end)
    orig = eng._front_end

    def checked(n, first, buf):
        assert eng._key_src is None or not (first <= eng._key_src < first + eng.W)
        orig(n, first, buf)
    eng._front_end = checked
    eng.stage_host(chunks[0], 0)
    eng.enqueue_staged(0, len(chunks[0]))
    buf = 0
    for k, chunk in enumerate(chunks):
        infos = eng.collect()
        if k + 1 < len(chunks):
            eng.stage_host(chunks[k + 1], 1 - buf)
            eng.enqueue_staged(1 - buf, len(chunks[k + 1]))
        if k % 2 == 0:
            eng.promote(infos[-1]["slot"])     # a keyframe that survives the next window: its half is refilled under it
        buf = 1 - buf
    assert any(ev[0] == "copy" for ev in eng.log)


def test_one_window_pending_and_argument_checks():
    eng = Recorder(3)
    with pytest.raises(RuntimeError):
        eng.collect()
    eng.stage_host([0, 1, 2], 0)
    eng.enqueue_staged(0, 3)
    with pytest.raises(RuntimeError):
        eng.enqueue_staged(1, 3)
    assert len(eng.collect()) == 3
    with pytest.raises(ValueError):
        eng.enqueue_staged(0, 4)
    assert eng.push_staged(0, 0) == []
    # a run that ends with a window still on the stream: reset() lets it finish before the buffers are reused
    eng.stage_host([7], 1)
    eng.enqueue_staged(1, 1)
    eng.reset()
    assert eng.ctx.synced == 1 and eng._pending is None
    eng.stage_host([0, 1, 2], 0)
    assert len(eng.push_staged(0, 3)) == 3
    # an upload issued ahead is consumed exactly once, and only when the frame count matches
    eng._up_pending[1] = 2
    eng._up_event = [_Ev(), _Ev()]
    eng.stage_host([3, 4], 1)
    eng.log = []
    eng.push_staged(1, 2)
    assert not any(ev[0] == "upload" for ev in eng.log) and eng._up_pending[1] == 0
    eng.stage_host([5, 6], 0)
    eng.push_staged(0, 2)
    assert any(ev[0] == "upload" for ev in eng.log)
