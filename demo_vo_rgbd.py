#!/usr/bin/env python3
"""Frame-to-frame visual odometry on an RGB-D sequence -- same entry point and arguments as the reference's
demo_vo_rgbd.py, running the hot path on the MI355X through libsosvo.

    python demo_vo_rgbd.py <sequence_path> --is_synthetic true [--hand_eye_transformation file] [--visualize_VO false]

<sequence_path>/rgbd/rgb/*.png and <sequence_path>/rgbd/depth/*.png (16-bit, millimetres) are the frames; results
go to <sequence_path>/results-rgbd/; <sequence_path>/rgbd/gt_TUM.txt is used as ground truth when present."""
import fnmatch
import os.path as osp
import sys
from argparse import ArgumentParser
from os import listdir

import numpy as np

ROOT = osp.dirname(osp.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main_rgbd_vo(argv=None):
    from vo_single_camera_sos_amd.omnistereo.common_tools import get_poses_from_file, make_sure_path_exists, str2bool
    parser = ArgumentParser(description="Demo of frame-to-frame visual odometry for RGB-D images.")
    parser.register("type", "bool", str2bool)
    parser.add_argument("sequence_path", nargs=1, help="The path to the sequence where the rgbd folder is located.")
    parser.add_argument("--is_synthetic", type="bool", default=True,
                        help="Synthetic data (radial depth, POV-Ray intrinsics) or real data (Z depth, 525 px)")
    parser.add_argument("--hand_eye_transformation", default="rgbd_hand_eye_transformation.txt", type=str)
    parser.add_argument("--visualize_VO", default=False, type="bool", help="not built, must stay false")
    parser.add_argument("--first_image_index", default=0, type=int)
    parser.add_argument("--last_image_index", default=-1, type=int)
    parser.add_argument("--step", default=1, type=int)
    parser.add_argument("--use_multithreads_for_VO", default=True, type="bool")
    parser.add_argument("--frame_window", default=-1, type=int,
                        help="sequence mode: frames per batched front-end pass on the GPU (-1 = default 32; 0 = the per-frame "
                             "mirror path).  The pose file does not depend on the window size.")
    args = parser.parse_args(argv)

    from vo_single_camera_sos_amd.omnistereo.camera_models import RGBDCamModel
    from vo_single_camera_sos_amd.omnistereo.pose_est_tools import driver_VO
    from vo_single_camera_sos_amd.omnistereo.transformations import rotation_matrix
    scene_path = osp.realpath(osp.expanduser(args.sequence_path[0]))
    hand_eye_T = None
    if args.is_synthetic:   # demo_vo_rgbd.py:66-84
        fx = fy = 554.256258
        depth_is_Z = False
        hand_eye_T = rotation_matrix(-np.pi / 2.0, [1, 0, 0])
    else:                   # :85-94
        fx = fy = 525.0
        depth_is_Z = True
        fn = osp.realpath(osp.expanduser(args.hand_eye_transformation))
        if osp.isfile(fn):
            hand_eye_T = get_poses_from_file(poses_filename=fn, input_units="m", output_working_units="m", indices=None,
                                             pose_format="tum", zero_up_wrt_origin=False, initial_T=None)[1][0]
    scene_path_rgbd = osp.join(scene_path, "rgbd")
    rgb_template = osp.join(scene_path_rgbd, "rgb", "*.png")
    depth_template = osp.join(scene_path_rgbd, "depth", "*.png")
    num_scene_images = len(fnmatch.filter(listdir(osp.join(scene_path_rgbd, "rgb")), "*.png"))
    results = osp.join(scene_path, "results-rgbd")
    make_sure_path_exists(results)
    _, scene_name = osp.split(scene_path)
    cam = RGBDCamModel(fx=fx, fy=fy, center_x=319.5, center_y=239.5, scaling_factor=1. / 1000.0, do_undistortion=False,
                       depth_is_Z=depth_is_Z, focal_length_m=1. / 1000.0)
    cam.T_Cest_wrt_Rgt = hand_eye_T
    out = driver_VO(camera_model=cam, scene_path=scene_path_rgbd, scene_path_vo_results=results,
                    scene_img_filename_template=rgb_template, depth_filename_template=depth_template,
                    num_scene_images=num_scene_images, visualize_VO=args.visualize_VO,
                    use_multithreads_for_VO=args.use_multithreads_for_VO, step_for_scene_images=args.step,
                    first_image_index=args.first_image_index, last_image_index=args.last_image_index,
                    thread_name="%s-%s" % (scene_name, "RGB-D"),
                    frame_window=None if args.frame_window < 0 else args.frame_window)
    print("GOODBYE!")
    return out


if __name__ == "__main__":
    main_rgbd_vo()
