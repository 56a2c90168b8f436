"""Fixture generator (build container only: needs /root/reference and PyYAML).

The reference holds seven scene-spec files (examples/*.yml, tests/data/*.yml; vocabulary of
pvtrace/cli/parse.py:83-466).  This script commits what they PARSE to -- `yaml.safe_load` of each, as one JSON
document `spec_dicts.json` -- plus the two data files the specs point at (a six-row CSV spectrum and an STL cube,
copied byte for byte into `spec_data/`), so that tests/test_spec.py can require `pvtrace_amd.spec.load` to build and
lower every one of them without the reference being present.

    python tests/golden/make_spec_fixtures.py
"""
import glob
import json
import os
import shutil

import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    for path in sorted(glob.glob(f"{REF}/examples/*.yml") + glob.glob(f"{REF}/tests/data/*.yml")):
        with open(path) as fp:
            out[os.path.relpath(path, REF)] = yaml.safe_load(fp)
    with open(os.path.join(HERE, "spec_dicts.json"), "w") as fp:
        json.dump(out, fp, indent=1, sort_keys=True)
    data = os.path.join(HERE, "spec_data")
    os.makedirs(os.path.join(data, "subfolder"), exist_ok=True)
    shutil.copyfile(f"{REF}/tests/data/subfolder/mock-spectrum.csv", os.path.join(data, "subfolder", "mock-spectrum.csv"))
    shutil.copyfile(f"{REF}/tests/data/20mm-xyz-cube.stl", os.path.join(data, "20mm-xyz-cube.stl"))
    os.chmod(os.path.join(data, "subfolder", "mock-spectrum.csv"), 0o644)
    os.chmod(os.path.join(data, "20mm-xyz-cube.stl"), 0o644)
    print(f"{len(out)} spec dicts ->", os.path.join(HERE, "spec_dicts.json"))


if __name__ == "__main__":
    main()
