"""GPU: error behaviour of the C ABI (include/sosvo.h: "every function returns SOSVO_OK or a negative sosvo_status;
the message of the last failure is kept per context").  Every compute entry point is called through raw ctypes
(i) with a NULL context and (ii) with a valid context but NULL buffers and unit sizes: it must come back with
SOSVO_ERR_ARG -- no launch, no crash -- and leave a message naming the function; the context stays usable."""
import ctypes

import numpy as np
import pytest
import torch

from vo_single_camera_sos_amd import _lib

pytestmark = pytest.mark.gpu

NOT_COMPUTE = {"sosvo_abi_version", "sosvo_orb_bit_pattern_31", "sosvo_create", "sosvo_destroy", "sosvo_set_stream", "sosvo_set_hint", "sosvo_synchronize",
               "sosvo_last_error", "sosvo_timer_start", "sosvo_timer_stop", "sosvo_timer_elapsed_ms", "sosvo_profile_enable",
               "sosvo_profile_count", "sosvo_profile_get", "sosvo_orb_pyramid_pixels", "sosvo_frame_pair_batch_workspace",
               "sosvo_rgbd_pair_batch_workspace", "sosvo_frame_pair_batch_streams_workspace", "sosvo_debug_fill_scratch",
               "sosvo_sequence_workspace", "sosvo_rgbd_sequence_workspace", "sosvo_frame_pair_batch_streams_join"}


def _dummy_args(argtypes, ctx_value):
    args = [ctx_value]
    for t in argtypes[1:]:
        if t is _lib.c_p or t is ctypes.c_char_p or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            args.append(None)
        elif t in (_lib.c_f64, _lib.c_f32):
            args.append(1.0)
        else:
            args.append(1)
    return args


def test_null_arguments_are_refused_not_dereferenced(ctx):
    lib = _lib.load()
    names = [n for n in _lib.SIGNATURES if n not in NOT_COMPUTE]
    assert len(names) >= 25
    for name in names:
        restype, argtypes = _lib.SIGNATURES[name]
        fn = getattr(lib, name)
        assert fn(*_dummy_args(argtypes, None)) == -1, name                      # NULL context
        rc = fn(*_dummy_args(argtypes, ctx._h))
        assert rc == -1, (name, rc)                                             # SOSVO_ERR_ARG
        msg = lib.sosvo_last_error(ctx._h).decode()
        assert name in msg and len(msg) > len(name) + 2, (name, msg)
    # workspace queries of a bad configuration answer 0 bytes
    assert lib.sosvo_frame_pair_batch_workspace(None) == 0 and lib.sosvo_rgbd_pair_batch_workspace(None) == 0
    assert lib.sosvo_frame_pair_batch_streams_workspace(None, 2) == 0 and lib.sosvo_sequence_workspace(None, 4, 9) == 0 \
        and lib.sosvo_rgbd_sequence_workspace(None, 4, 9) == 0
    assert lib.sosvo_set_hint(None, 1, 1) == -1 and lib.sosvo_set_hint(ctx._h, 99, 1) == -1 and lib.sosvo_set_hint(ctx._h, 1, 0) == 0
    assert lib.sosvo_frame_pair_batch_streams_join(None) == -1 and lib.sosvo_frame_pair_batch_streams_join(ctx._h) == 0
    # the context is still good for real work
    q = torch.randint(0, 256, (1, 8, 32), dtype=torch.uint8, device=ctx.device)
    n = torch.tensor([8], dtype=torch.int32, device=ctx.device)
    keys = ctx.match_hamming(q, q, n, n, k=1)
    ctx.synchronize()
    assert np.array_equal(keys.cpu().numpy()[0, :, 0] & _lib.KEY_IDX_MASK, np.arange(8))
