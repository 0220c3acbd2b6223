"""Host-side counterpart of the reference's camera thread (omnistereo/webcam_live.py:256-291, CamAsWorkingThread):
a thread that keeps polling a frame source and holds the most recent frame for the live VO driver
(pose_est_tools.run_VO_live / driver_VO_live).  The reference wraps a cv2.VideoCapture with GUI sliders; here the
source is any object with `get_single_frame() -> (success, frame)` -- a camera wrapper, a directory of images
(ImageSequenceCam), or a test's generator -- and there is no window."""
import threading
import time


class FrameSourceThread(threading.Thread):
    """current_frame: the newest frame (None before the first one and after the source ends); quit_flag: set by the
    driver to stop polling.  lockstep=True (for replays and tests): reading current_frame hands out every frame exactly
    once -- the read waits for a frame that has not been handed out yet, and the source is polled again only after
    that read -- instead of the free-running behaviour of a camera."""

    def __init__(self, cam, min_period_s=0.0, lockstep=False):
        threading.Thread.__init__(self)
        self.cam = cam
        if not hasattr(cam, "show_img"):
            cam.show_img = False
        self.min_period_s, self.lockstep = float(min_period_s), bool(lockstep)
        self._frame, self._fresh, self._ended = None, False, False
        self._cv = threading.Condition()
        self.quit_flag = False
        self.frames_delivered = 0

    @property
    def current_frame(self):
        with self._cv:
            if self.lockstep:
                while not self._fresh and not self._ended and not self.quit_flag and (self.is_alive() or self.ident is None):
                    self._cv.wait(0.05)
                    if self.ident is None:   # not started yet: nothing to wait for
                        break
                frame = self._frame if self._fresh else None
                self._fresh = False
                self._cv.notify_all()
                return frame
            return self._frame

    def run(self):
        while not self.quit_flag:
            t0 = time.perf_counter()
            success, frame = self.cam.get_single_frame()
            if not success:
                break
            with self._cv:
                self._frame, self._fresh = frame, True
                self.frames_delivered += 1
                self._cv.notify_all()
                while self.lockstep and self._fresh and not self.quit_flag:
                    self._cv.wait(0.05)
            dt = self.min_period_s - (time.perf_counter() - t0)
            if dt > 0:
                time.sleep(dt)
        with self._cv:
            self._frame, self._fresh, self._ended = None, False, True
            self._cv.notify_all()


class ImageSequenceCam(object):
    """A 'camera' that replays image files in sorted order (one per get_single_frame call)."""

    def __init__(self, filename_template, show_img=False):
        from .common_cv import get_images
        self.names = get_images(filename_template, indices_list=None, return_names_only=True)
        self.k = 0
        self.show_img = show_img

    def get_single_frame(self):
        from .common_cv import imread
        if self.k >= len(self.names):
            return False, None
        img = imread(self.names[self.k])
        self.k += 1
        return True, img
