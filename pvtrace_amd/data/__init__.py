"""Closed-form dye spectra shipped with the scene API (data, not code paths)."""
from pvtrace_amd.data import fluro_red, lumogen_f_red_305  # noqa: F401
