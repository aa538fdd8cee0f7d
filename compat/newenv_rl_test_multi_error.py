"""Import shim: put this directory on PYTHONPATH ahead of the reference checkout and
DOODLE's scripts get the MI355X HelioField."""
from doodle_amd.field import HelioField  # noqa: F401
from doodle_amd.optics_functions import (  # noqa: F401  (README.md:203-207, "internals you may reuse")
    gaussian_blur_batch, ray_plane_intersection_batch, reflect_vectors, rotate_normals_batch)
