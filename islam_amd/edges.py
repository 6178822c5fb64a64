"""Edge mask of TartanVO.forward on the device (reference TartanVO.py:145-155).

The reference copies img0 to the host (27.5 MB D2H at B=8), loops over the batch with
cv2.resize(1/4) -> cv2.Canny(50, 100) -> cv2.dilate(5x5) and copies the mask back: a full device
synchronisation in the middle of the forward.  Here the whole pipeline is ONE HIP launch
(islam_edge_mask, islam_amd/csrc/edge_mask.hip: one workgroup per image, the quarter-resolution image,
gradient magnitudes and the hysteresis map in LDS, hysteresis iterated to its fixed point inside the
kernel) -- no host round trip, no synchronisation.  The integer arithmetic is OpenCV 4.7's
(restated for the tests in oracle/canny.py; opencv is not installable in the build container, so this
piece is "parity unpinned").
"""
from . import ops


def edge_mask(img0, downscale=True):
    """img0: (B,3,H,W) float in [0,1] (BGR/255 like the reference) -> (B,H/4,W/4) bool."""
    return ops.edge_mask(img0, downscale=downscale, low=50, high=100)
