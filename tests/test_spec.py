"""Scene-spec front-end (YAML / dict): the reference's spec vocabulary
(pvtrace/cli/parse.py; tests/test_engine.py:353-415 for the recorder parts)."""
import numpy as np
import pytest

from pvtrace_amd import spec
from pvtrace_amd.engine import UnsupportedSceneError, compile_scene
from pvtrace_amd.light import ConstantWavelengthMask, RectangularMask
from pvtrace_amd.material import Cone, HenyeyGreenstein, Luminophore

LSC_YAML = """
version: "1.0"
nodes:
  world:
    sphere:
      radius: 12.0
      material:
        refractive-index: 1.0
  lsc:
    parent: world
    location: [0, 0, 0.5]
    record: true
    box:
      size: [5, 5, 1]
      material:
        refractive-index: 1.5
        components: [dye, background]
  laser:
    parent: world
    location: [0, 0, 3.0]
    direction: [0, 0, -1]
    light:
      wavelength: 555
      mask:
        direction:
          cone:
            half-angle: 22.5
components:
  dye:
    luminophore:
      absorption:
        coefficient: 5
        spectrum: {name: lumogen-f-red-305, range: {min: 500, max: 1000, spacing: 2}}
      emission:
        quantum-yield: 0.95
        phase-function: isotropic
        spectrum: {name: lumogen-f-red-305, range: {min: 500, max: 1000, spacing: 2}}
  background:
    absorber:
      coefficient: 0.05
recorders:
  edge-escape:
    node: lsc
    event: escaping
    facet: [1, 0, 0]
    histograms:
      wavelength: [500, 900, 80]
      angle: [0, 1.5708, 18]
      duration: [0, 1.0e-9, 60]
  lsc-top:
    node: lsc
    event: escaping
    facet: [0, 0, 1]
"""


def recorders_of(scene):
    return {r.name: r for n in scene.root.preorder() for r in n.recorders}


def test_yaml_scene_with_auto_and_explicit_recorders():
    scene = spec.load(LSC_YAML)
    recs = recorders_of(scene)
    assert recs["edge-escape"].facet == (1.0, 0.0, 0.0) and len(recs["edge-escape"].histograms) == 3
    auto = {"lsc-top", "lsc-bottom", "lsc-east", "lsc-west", "lsc-north", "lsc-south", "lsc-lost"}
    assert auto <= set(recs)
    assert len(recs["lsc-top"].histograms) == 0          # the explicit entry wins over the shorthand
    assert len(recs["lsc-bottom"].histograms) == 3
    c = compile_scene(scene)
    assert c.node_names == ["world", "lsc"] and c.geom_type.tolist() == [1, 0]
    assert c.local_to_world[1][:3, 3].tolist() == [0.0, 0.0, 0.5]
    assert c.comp_type.tolist() == [2, 0] and c.comp_qy[0] == 0.95
    assert c.comp_abs_n[0] == 251 and np.isclose(c.abs_y[:251].max(), 5.0)
    laser = scene.light_nodes[0]
    assert isinstance(laser.light.wavelength, ConstantWavelengthMask) and laser.light.wavelength() == 555.0
    assert isinstance(laser.light.direction, Cone) and np.isclose(laser.light.direction.theta_max, np.radians(22.5))
    # `direction: [0, 0, -1]` points the light down
    assert np.allclose(laser.vector_to_node((0, 0, 1), scene.root), (0, 0, -1))


def test_dict_spec_masks_phase_functions_and_errors(tmp_path):
    csv = tmp_path / "abs.csv"
    x = np.linspace(400, 700, 31)
    csv.write_text("i,x,y\n" + "\n".join(f"{i},{a},{np.exp(-((a - 550) / 40) ** 2)}" for i, a in enumerate(x)))
    base = {
        "version": "1.0",
        "nodes": {
            "world": {"box": {"size": [20, 20, 20], "material": {"refractive-index": 1.0}}},
            "rod": {"location": [1, 0, 0], "cylinder": {"length": 4, "radius": 0.5, "material": {
                "refractive-index": 1.6, "components": ["mist", "stain"]}}},
            "lamp": {"location": [0, 0, 5], "direction": [0, 0, -1], "light": {"mask": {
                "wavelength": {"nanometers": 610}, "position": {"rect": [1.0, 2.0]},
                "direction": {"henyey-greenstein": {"g": 0.8}}}}},
        },
        "components": {
            "mist": {"scatterer": {"coefficient": 0.7, "quantum-yield": 0.9,
                                   "phase-function": {"cone": {"half-angle": 10}}}},
            "stain": {"absorber": {"coefficient": 2.0, "spectrum": {"file": str(csv)}}},
        },
    }
    scene = spec.load(base)
    c = compile_scene(scene)
    assert c.geom_type.tolist() == [0, 2] and c.comp_type.tolist() == [1, 0]
    assert c.comp_phase_type[0] == 2 and np.isclose(c.comp_phase_param[0], np.radians(10))
    assert c.comp_abs_n.tolist() == [1, 31] and np.isclose(c.abs_y.max(), 2.0)
    lamp = scene.light_nodes[0].light
    assert isinstance(lamp.position, RectangularMask) and isinstance(lamp.direction, HenyeyGreenstein)
    bad = dict(base, nodes=dict(base["nodes"], blob={"mesh": {"file": "x.obj", "material": {"refractive-index": 1.5}}}))
    with pytest.raises(UnsupportedSceneError):
        spec.load(bad)
    from pvtrace_amd import mesh as M
    M.save_stl(str(tmp_path / "gem.stl"), *M.icosphere(1, 0.4))
    meshed = spec.load(dict(base, nodes=dict(base["nodes"], gem={"mesh": {
        "file": str(tmp_path / "gem.stl"), "material": {"refractive-index": 1.5}}, "location": [0, 0, 3]})))
    cm = compile_scene(meshed)
    assert cm.geom_type.tolist() == [0, 2, 3] and cm.n_mesh_faces == 80 and cm.mesh_face_count.tolist() == [0, 0, 80]
    with pytest.raises(spec.SpecError):
        spec.load({"version": "9.9", "nodes": {}})
    with pytest.raises(spec.SpecError):
        spec.load(dict(base, nodes=dict(base["nodes"], rod={"cylinder": {"length": 1, "radius": 1, "material": {
            "refractive-index": 1.5, "components": ["nope"]}}})))
    with pytest.raises(ValueError):
        spec.load(dict(base, recorders={"r": {"node": "ghost", "event": "entering"}}))


@pytest.mark.gpu
def test_spec_scene_traces_on_the_engine():
    """reference tests/test_engine.py:376-415: the shorthand's recorders collect photons."""
    from pvtrace_amd import engine

    scene = spec.load(LSC_YAML)
    result = engine.simulate(scene, 20000, seed=2, record_every=0, emit_seed=1)
    recs = result.recorders
    assert recs["lsc-top"].rays > 0 and recs["edge-escape"].rays > 0 and recs["lsc-lost"].rays > 0
    assert recs["edge-escape"].rays == recs["lsc-east"].rays    # same facet, explicit vs shorthand
    edges, values = recs["edge-escape"].histogram(0)
    assert values.sum() == recs["edge-escape"].rays


def test_every_scene_spec_the_reference_holds_loads_and_lowers():
    """tests/golden/spec_dicts.json = yaml.safe_load of the reference's seven spec files (examples/*.yml,
    tests/data/*.yml; tests/golden/make_spec_fixtures.py), with the CSV spectrum and the STL cube they point at.  Each
    must build, lower to tables and survive a few hundred photons on the referee.  The parser fixture
    (tests/data/pvtrace-scene-spec.yml; reference tests/test_cli.py) is the one with every mask, every phase function
    -- `lambertian` among them, cli/parse.py:166-167 -- and a mesh."""
    import json
    import os

    from oracle import oracle as O
    from pvtrace_amd.engine.compiler import PHASE_LAMBERTIAN
    from pvtrace_amd.engine.emit import emit_bundle
    from tests.util import GOLD

    with open(os.path.join(GOLD, "spec_dicts.json")) as fp:
        docs = json.load(fp)
    assert len(docs) == 7
    for name, doc in docs.items():
        scene = spec.load(doc, base=os.path.join(GOLD, "spec_data"))
        compiled = compile_scene(scene)
        assert len(scene.light_nodes) >= 1, name
        pos, dirs, wl, _ = emit_bundle(scene, 300, seed=4)
        out = O.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 64, 0, 1, 1, math_mode=O.MATH_PORTABLE)
        assert out["counts"].min() >= 2, name
    big = compile_scene(spec.load(docs["tests/data/pvtrace-scene-spec.yml"], base=os.path.join(GOLD, "spec_data")))
    assert sorted(set(big.comp_phase_type.tolist())) == [0, 1, 2, PHASE_LAMBERTIAN]
    assert big.node_names == ["world", "ball", "rod", "xyz-cube"] and big.geom_type.tolist() == [0, 1, 2, 3]
    assert len(big.scene.light_nodes) == 8
