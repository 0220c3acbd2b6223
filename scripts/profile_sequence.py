"""cProfile of run_VO in sequence mode (frame_window 32) on the bench's synthetic sequence: where the host spends a frame.
    python scripts/profile_sequence.py [--frames 256] [--top 35]"""
import argparse
import contextlib
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--top", type=int, default=35)
    ap.add_argument("--window", type=int, default=32)
    ap.add_argument("--stage-threads", type=int, default=0, help="threads of the staging copies (0 = the package's default)")
    ap.add_argument("--ab", type=int, default=0, help="A/B of the early upload (copy stream) and of enqueueing the next window ahead: "
                                                     "this many runs of each mode, interleaved; medians")
    args = ap.parse_args()
    from vo_single_camera_sos_amd import synthetic
    from vo_single_camera_sos_amd.omnistereo.gum import synthetic_gums
    from vo_single_camera_sos_amd.omnistereo.panorama import Panorama
    gs = synthetic_gums()
    for m in (gs.top_model, gs.bot_model):
        m.panorama = Panorama(m, width=1440)
    gs.make_annulus_masks((480, 640))
    seq, _ = synthetic.make_sequence(gs, args.frames, seed=1241, workers=min(16, len(os.sched_getaffinity(0))))
    import torch
    from vo_single_camera_sos_amd.omnistereo.pose_est_tools import run_VO
    if args.stage_threads > 0:
        from vo_single_camera_sos_amd import pipeline as _p
        _p._STAGE_THREADS = args.stage_threads

    def frames(n=args.frames):
        for k in range(n):
            yield k, seq[k], None
    if args.ab > 0:
        from vo_single_camera_sos_amd import pipeline
        times = {0: [], 1: [], 2: []}
        with tempfile.TemporaryDirectory() as d:
            with contextlib.redirect_stdout(io.StringIO()):
                run_VO(None, gs, results_path=d, _live_frames=lambda: frames(min(args.frames, 34)), frame_window=args.window)
                for k in range(3 * args.ab):
                    on = k % 3          # 0: neither, 1: early upload, 2: early upload + the next window enqueued ahead
                    pipeline._SequenceBase.early_upload = on >= 1
                    pipeline._SequenceBase.enqueue_ahead = on >= 2
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    run_VO(None, gs, results_path=d, _live_frames=frames, frame_window=args.window)
                    torch.cuda.synchronize()
                    times[on].append(time.perf_counter() - t0)
        for on in (0, 1, 2):
            ts = sorted(times[on])
            print("mode %d (0 neither, 1 early upload, 2 + enqueue ahead): median %.2f ms per %d frames = %.0f frames/s (min %.2f, max %.2f ms)" % (
                on, 1e3 * ts[len(ts) // 2], args.frames, args.frames / ts[len(ts) // 2], 1e3 * ts[0], 1e3 * ts[-1]))
        return
    with tempfile.TemporaryDirectory() as d:
        with contextlib.redirect_stdout(io.StringIO()):
            run_VO(None, gs, results_path=d, _live_frames=lambda: frames(min(args.frames, 34)), frame_window=args.window)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = run_VO(None, gs, results_path=d, _live_frames=frames, frame_window=args.window)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            pr = cProfile.Profile()
            pr.enable()
            run_VO(None, gs, results_path=d, _live_frames=frames, frame_window=args.window)
            torch.cuda.synchronize()
            pr.disable()
    print("unprofiled: %.1f frames/s, %.3f ms per frame; stages %s" % (args.frames / dt, 1e3 * dt / args.frames, r["sequence_mode"]["stage_s"]))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(args.top)
    print(s.getvalue())


if __name__ == "__main__":
    main()
