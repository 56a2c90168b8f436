"""pytest plugin (`-p tests.ref_compat_plugin`): makes `pvtrace_amd` answer to the name `pvtrace` before any test module is
imported -- used by tests/test_reference_unit_tests.py to run the reference's OWN unit-test files against this package."""
import sys

sys.dont_write_bytecode = True   # nothing is written next to the reference's files

import pvtrace_amd.compat  # noqa: E402

pvtrace_amd.compat.install()
