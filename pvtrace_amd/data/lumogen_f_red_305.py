"""Lumogen F Red 305 spectra as Gaussian fits.

Same fit parameters and evaluation order as the reference's
pvtrace/data/lumogen_f_red_305.py:4-75 (values pinned bit-for-bit by
tests/golden/spectra.npz, generated from the reference module itself).
"""
import numpy as np

# (amplitude, centre nm, width nm) of the absorption fit, in summation order
_ABSORPTION_TERMS = (
    (0.9454846839252642, 578.6167306868869, 22.69760939870020),
    (0.6430326869158796, 535.1850303736512, 28.63029894331116),
    (0.1243340609168971, 494.5721783546976, 13.98438275367119),
    (0.3651471532322375, 440.4679754085741, 34.91923613222621),
    (0.7042787252835550, 336.0548556730901, 34.24136755250487),
)
_EMISSION_TERM = (1.0, 600.0, 38.60)


def _term(x, amp, centre, width):
    return amp * np.exp(-(((centre - x) / width) ** 2))


def absorption(x):
    """Absorption line shape on wavelengths `x` (nm), peak-normalised to 1."""
    total = _term(x, *_ABSORPTION_TERMS[0])
    for params in _ABSORPTION_TERMS[1:]:
        total = total + _term(x, *params)
    return total / np.max(total)


def emission(x):
    """Emission line shape on wavelengths `x` (nm), peak value 1."""
    return _term(x, *_EMISSION_TERM)
