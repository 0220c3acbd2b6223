#!/usr/bin/env python3
"""Live frame-to-frame visual odometry on the single-camera omnistereo (SOS) rig -- the entry point and arguments of the
reference's demo_vo_sos_live.py (:47-109), running the hot path on the MI355X through libsosvo.

    python demo_vo_sos_live.py <results_path> --calibrated_gums_file gums-calibrated.json [--frames 'dir/image-*.png']

The reference opens camera 0 through its OpenCV wrapper (WebcamLive, a GUI class that is out of scope here) and hands
a camera thread to driver_VO_live.  This entry point wraps the SAME driver over any frame source: by default
`--frames` names recorded omni frames that are replayed as a camera would deliver them (ImageSequenceCam); a host
application passes its own object with get_single_frame() to main_sos_vo_live(frame_source=...).
Results go to <results_path>/results-omni/ (estimated_frame_poses_TUM.txt, keyframe_ids.txt, printed_messages.log)."""
import os.path as osp
import sys
from argparse import ArgumentParser

ROOT = osp.dirname(osp.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main_sos_vo_live(argv=None, frame_source=None):
    from vo_single_camera_sos_amd.omnistereo.common_tools import make_sure_path_exists, str2bool
    parser = ArgumentParser(description="Live demo of frame-to-frame visual odometry for the Single-camera SOS rig.")
    parser.register("type", "bool", str2bool)
    parser.add_argument("results_path", nargs=1, help="The path where the results will be saved into.")
    parser.add_argument("--calibrated_gums_file", default="gums-calibrated.json", type=str,
                        help="Complete path and name of the calibrated GUMS file (JSON)")
    parser.add_argument("--visualize_VO", default=False, type="bool",
                        help="(Optional) 3-D visualisation of the trajectory: not built, must stay false")
    parser.add_argument("--frames", default=None, type=str,
                        help="glob of recorded omni frames replayed as the camera (needed unless a frame_source is passed in)")
    parser.add_argument("--lockstep", default=True, type="bool",
                        help="deliver every frame exactly once (reproducible runs); false = newest frame wins, as a camera")
    parser.add_argument("--use_multithreads_for_VO", default=True, type="bool")
    parser.add_argument("--frame_window", default=-1, type=int,
                        help="frames per batched front-end pass (-1 = 1 for a live source; 0 = the per-frame mirror path)")
    args = parser.parse_args(argv)

    from vo_single_camera_sos_amd.omnistereo.gum import load_gums_json
    from vo_single_camera_sos_amd.omnistereo.pose_est_tools import driver_VO_live
    from vo_single_camera_sos_amd.omnistereo.webcam_live import FrameSourceThread, ImageSequenceCam
    scene_path = osp.realpath(osp.expanduser(args.results_path[0]))
    results = osp.join(scene_path, "results-omni")
    make_sure_path_exists(results)
    if frame_source is None:
        if not args.frames:
            parser.error("no camera wrapper is built in: give --frames <glob> or call main_sos_vo_live(frame_source=...)")
        frame_source = ImageSequenceCam(osp.expanduser(args.frames))
    cam_working_thread = FrameSourceThread(frame_source, lockstep=args.lockstep)
    gums_calibrated = load_gums_json(osp.realpath(osp.expanduser(args.calibrated_gums_file)))
    out = driver_VO_live(gums_calibrated, results, cam_working_thread, visualize_VO=args.visualize_VO,
                         use_multithreads_for_VO=args.use_multithreads_for_VO, thread_name="LIVE-SOS",
                         frame_window=None if args.frame_window < 0 else args.frame_window)
    print("GOODBYE!")
    return out


if __name__ == "__main__":
    main_sos_vo_live()
