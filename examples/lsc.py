"""A 5 x 5 x 1 cm Lumogen F Red luminescent solar concentrator with solar cells on two edges and
a back-surface mirror, through the high-level `LSC` builder (reference pvtrace/device/lsc.py):
counts per face split into solar / luminescent light come from source-filtered recorders on the GPU
instead of a pandas table of every ray.

    python examples/lsc.py                    # needs an MI355X
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pvtrace_amd import LSC                                  # noqa: E402
from pvtrace_amd.data import lumogen_f_red_305               # noqa: E402
from pvtrace_amd.light import RectangularMask                # noqa: E402

x = np.arange(400, 800, dtype=float)
lsc = LSC((5.0, 5.0, 1.0), wavelength_range=x, n1=1.5)
lsc.add_luminophore("Lumogen F Red 305", np.column_stack((x, lumogen_f_red_305.absorption(x) * 10.0)),
                    np.column_stack((x, lumogen_f_red_305.emission(x))), quantum_yield=0.98)
lsc.add_absorber("PMMA", 0.02)
lsc.add_light("top illumination", (0.0, 0.0, 0.6), rotation=(np.radians(180), (1, 0, 0)),
              position=RectangularMask(2.5, 2.5))
lsc.add_solar_cell({"left", "right"})
lsc.add_back_surface_mirror()
lsc.simulate(2_000_000, seed=3)
lsc.report()
