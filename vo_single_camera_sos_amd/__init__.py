"""vo_single_camera_sos_amd -- MI355X-native front end of the single-camera SOS visual
odometry hot path (unwrap -> detect/describe -> Hamming match -> bearings ->
triangulation -> absolute-pose RANSAC).  The arithmetic lives in libsosvo.so (HIP,
gfx950, C ABI in include/sosvo.h); this package holds the ctypes binding and the
host-side mirror of the reference's `omnistereo` interfaces for that path only."""

__all__ = ["_lib", "device"]
