"""vo_single_camera_sos_amd -- MI355X-native front end of the single-camera SOS visual
odometry hot path (unwrap -> detect/describe -> Hamming match -> bearings ->
triangulation -> absolute-pose RANSAC).  The arithmetic lives in libsosvo.so (HIP,
gfx950, C ABI in include/sosvo.h); this package holds the ctypes binding and the
host-side mirror of the reference's `omnistereo` interfaces for that path only."""

import os as _os

# The engines keep 4 - 10 HIP streams busy (three parts of a batch, copy streams of the host-fed form, the library's own
# streams behind the one-call entry).  The ROCm runtime multiplexes a process's streams over GPU_MAX_HW_QUEUES hardware
# queues, and streams that share one are serialised: with the runtime's default the host-fed step reaches 26 k pairs/s and
# the one-call C entry falls to its join-every-call rate whenever other streams of the process are alive; with 12 or more
# queues they reach 30 k and 99 % of the engine (measured, bench.py); a long-lived process that has created several engines
# (torch hands out its 32 pooled streams in turn) still loses ~10 % on a later engine at 16 and nothing at 32.  Read by the HIP runtime when it initialises, i.e. at the
# first GPU call after this import; an explicit setting of the variable wins.  A C host sets it the same way before its
# first HIP call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

__all__ = ["_lib", "device"]
