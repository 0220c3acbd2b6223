#!/usr/bin/env python3
"""Runs on the REFERENCE's side (where `omnistereo` of ubuntuslave/vo_single_camera_sos and its cv2 import): turns the
calibrated rig a user of the reference owns -- the pickle of a live GUMStereo object, `gums-calibrated.pkl`
(demo_vo_sos.py:57, :109; omnistereo/common_tools.py:131-138) -- into the `sosvo-gums-1` JSON document this repository
reads (vo_single_camera_sos_amd/omnistereo/gum.py: load_gums_json).  Only attribute access, json and numpy: the function
works on any object that carries the reference's attribute names (gum.py:77-116, :169-214 for the parameters;
camera_models.py:884-936, :1384-1480 for limits, radii and centres), which is how tests/test_host_vo_helpers.py runs it.

    PYTHONPATH=/path/to/vo_single_camera_sos python export_reference_gums.py gums-calibrated.pkl rig.json
"""
import json
import sys

import numpy as np


def gums_to_sosvo_json(gums, panorama_width=None):
    """GUMStereo-like object -> dict in the `sosvo-gums-1` layout."""
    def pt(v):
        return [float(x) for x in np.ravel(v)[:2]]

    def mirror(m):
        p = m.precalib_params
        cp = np.ravel(getattr(m, "Cp_wrt_M", None) if getattr(m, "Cp_wrt_M", None) is not None else [p.xi1, p.xi2, p.xi3])
        centre = pt(getattr(p, "center_point", (p.u_center, p.v_center)))
        inner = getattr(p, "center_point_inner", None)
        outer = getattr(p, "center_point_outer", None)
        T = np.asarray(getattr(m, "T_model_wrt_C", np.identity(4)), dtype=np.float64)
        d = dict(xi1=float(cp[0]), xi2=float(cp[1]), xi3=float(cp[2]), k1=float(p.k1), k2=float(p.k2), k3=float(p.k3),
                 gamma1=float(p.gamma1), gamma2=float(p.gamma2), alpha_c=float(p.alpha_c), u_center=float(p.u_center),
                 v_center=float(p.v_center), use_distortion=bool(getattr(p, "use_distortion", True)), z_axis=float(m.z_axis),
                 F=[float(v) for v in np.ravel(m.F)[:3]], lowest_elevation_angle=float(m.lowest_elevation_angle),
                 highest_elevation_angle=float(m.highest_elevation_angle), inner_img_radius=float(m.inner_img_radius),
                 outer_img_radius=float(m.outer_img_radius), center_point=centre,
                 center_point_inner=centre if inner is None else pt(inner), center_point_outer=centre if outer is None else pt(outer),
                 image_size=[int(v) for v in np.ravel(p.image_size)[:2]])
        if not np.array_equal(T[:3, :3], np.identity(3)):
            d["R_model_wrt_C"] = [[float(v) for v in row] for row in T[:3, :3]]
        return d
    top, bot = gums.top_model, gums.bot_model
    width = panorama_width
    if width is None:
        pano = getattr(top, "panorama", None)
        width = int(pano.cols) if pano is not None else 1200
    return dict(format="sosvo-gums-1", units=str(getattr(gums, "units", "mm")), panorama_width=int(width), top=mirror(top),
                bottom=mirror(bot))


def main(argv):
    if len(argv) != 3:
        sys.exit(__doc__)
    from omnistereo.common_tools import load_obj_from_pickle   # the reference's own loader
    doc = gums_to_sosvo_json(load_obj_from_pickle(argv[1]))
    with open(argv[2], "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", argv[2])


if __name__ == "__main__":
    main(sys.argv)
