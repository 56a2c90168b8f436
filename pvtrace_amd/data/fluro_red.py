"""Fluro Red spectra: four-Gaussian absorption, exponentially-modified-Gaussian
emission.  Same fit parameters and evaluation order as the reference's
pvtrace/data/fluro_red.py (pinned by tests/golden/spectra.npz)."""
import numpy as np
from scipy.special import erf

# (centre nm, amplitude, width nm) in summation order
_ABSORPTION_TERMS = (
    (549.06438843562137, 439.06754804626956, 24.298601639828647),
    (379.48645797468572, 85.177292848284353, 13.513987279089216),
    (519.58858977131513, 660.1731296017241, 38.263352007649125),
    (490.05625608592726, 511.11501615291041, 52.213294432464529),
)
_EMG = (1.1477763237584664, 592.06478874548839, 19.981040318195117, 12.723704058786568)


def absorption(x):
    total = None
    for centre, amp, width in _ABSORPTION_TERMS:
        part = amp * np.exp(-(((centre - x) / width) ** 2))
        total = part if total is None else total + part
    return total / np.max(total)


def emission(x):
    a, b, c, d = _EMG
    r2 = np.sqrt(2)
    return (
        a * c * np.sqrt(2 * np.pi) / (2 * d)
        * np.exp((c ** 2 / (2 * d ** 2)) - ((x - b) / d))
        * (d / np.abs(d) + erf((x - b) / (r2 * c) - c / (r2 * d)))
    )
