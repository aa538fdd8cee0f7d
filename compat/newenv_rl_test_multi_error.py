"""Import shim: put this directory on PYTHONPATH ahead of the reference checkout and
DOODLE's scripts get the MI355X HelioField."""
from doodle_amd.field import HelioField  # noqa: F401
