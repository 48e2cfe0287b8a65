"""Data edges of the hot path: the arithmetic of the reference's dataset classes that builds the `target` dict the
tick_* functions consume.  Image decoding / resizing (imageio, cv2 in the reference) stays with the caller: these functions take the
decoded arrays."""
from .targets import get_ndc_matrix_from_ss, camera_matrices, make_target  # noqa: F401
