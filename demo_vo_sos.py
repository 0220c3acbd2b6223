#!/usr/bin/env python3
"""Frame-to-frame visual odometry on a single-camera omnistereo (SOS) sequence -- same entry point and arguments
as the reference's demo_vo_sos.py, running the hot path on the MI355X through libsosvo.

    python demo_vo_sos.py <sequence_path> --calibrated_gums_file gums-calibrated.json [--visualize_VO false]

<sequence_path>/omni/image-*.png are the omni frames; results go to <sequence_path>/results-omni/
(estimated_frame_poses_TUM.txt, gt_associated_frame_poses_TUM.txt, keyframe_ids.txt, printed_messages.log);
<sequence_path>/omni/gt_TUM.txt is used as ground truth when present.  The calibrated rig is a JSON document
(vo_single_camera_sos_amd.omnistereo.gum.save_gums_json) instead of the reference's pickle of live objects."""
import fnmatch
import os.path as osp
import sys
from argparse import ArgumentParser
from os import listdir

ROOT = osp.dirname(osp.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main_sos_vo(argv=None):
    from vo_single_camera_sos_amd.omnistereo.common_tools import make_sure_path_exists, str2bool
    parser = ArgumentParser(description="Demo of frame-to-frame visual odometry for the Single-camera SOS images.")
    parser.register("type", "bool", str2bool)
    parser.add_argument("sequence_path", nargs=1, help="The path to the sequence where the omni folder is located.")
    parser.add_argument("--calibrated_gums_file", default="gums-calibrated.json", type=str,
                        help="Complete path and name of the calibrated GUMS file (JSON)")
    parser.add_argument("--visualize_VO", default=False, type="bool",
                        help="(Optional) 3-D visualisation of the trajectory: not built, must stay false")
    parser.add_argument("--first_image_index", default=0, type=int)
    parser.add_argument("--last_image_index", default=-1, type=int, help="-1 for up to the last one")
    parser.add_argument("--step", default=1, type=int)
    parser.add_argument("--use_multithreads_for_VO", default=True, type="bool")
    parser.add_argument("--frame_window", default=-1, type=int,
                        help="sequence mode: frames per batched front-end pass on the GPU (-1 = default 32; 0 = the per-frame "
                             "mirror path).  The pose file does not depend on the window size.")
    args = parser.parse_args(argv)

    from vo_single_camera_sos_amd.omnistereo.gum import load_gums_json
    from vo_single_camera_sos_amd.omnistereo.pose_est_tools import driver_VO
    scene_path = osp.realpath(osp.expanduser(args.sequence_path[0]))
    gums_file = osp.realpath(osp.expanduser(args.calibrated_gums_file))
    scene_prefix_filename = "image-*.png"
    scene_path_omni = osp.join(scene_path, "omni")
    template = osp.join(scene_path_omni, scene_prefix_filename)
    num_scene_images = len(fnmatch.filter(listdir(scene_path_omni), scene_prefix_filename))
    results = osp.join(scene_path, "results-omni")
    make_sure_path_exists(results)
    _, scene_name = osp.split(scene_path)
    gums_calibrated = load_gums_json(gums_file)
    out = driver_VO(camera_model=gums_calibrated, scene_path=scene_path_omni, scene_path_vo_results=results,
                    scene_img_filename_template=template, depth_filename_template=None, num_scene_images=num_scene_images,
                    visualize_VO=args.visualize_VO, use_multithreads_for_VO=args.use_multithreads_for_VO,
                    step_for_scene_images=args.step, first_image_index=args.first_image_index,
                    last_image_index=args.last_image_index, thread_name="%s-%s" % (scene_name, "SOS"),
                    frame_window=None if args.frame_window < 0 else args.frame_window)
    print("GOODBYE!")
    return out


if __name__ == "__main__":
    main_sos_vo()
