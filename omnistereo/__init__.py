"""The reference's module paths for the VO path: `import omnistereo.pose_est_tools`, `from omnistereo.camera_models import
FeatureMatcher`, ... resolve to the MI355X-backed mirror (vo_single_camera_sos_amd.omnistereo) -- the SAME module objects,
not copies -- whenever the repository root is on sys.path, as it is for demo_vo_sos.py / demo_vo_rgbd.py.  Nothing of the
reference is here: the modules hold only what the frame-to-frame VO path needs."""
import importlib
import sys

_MODULES = ("transformations", "common_tools", "common_cv", "gum", "panorama", "camera_models", "pose_est_tools", "webcam_live")
for _n in _MODULES:
    _m = importlib.import_module("vo_single_camera_sos_amd.omnistereo." + _n)
    sys.modules[__name__ + "." + _n] = _m
    globals()[_n] = _m
del _n, _m
