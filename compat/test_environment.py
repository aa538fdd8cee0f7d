"""Import shim for DOODLE's `from test_environment import HelioEnv`."""
from doodle_amd.env import (  # noqa: F401
    HelioEnv, azimuth_elevation_to_primary_direction, make_distance_maps, sample_cone_directions)
