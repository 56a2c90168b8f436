"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Nothing of the reference is copied: its modules are imported from where they
lie and only numeric inputs/outputs are saved.

* The reference's native kernel (pvtrace/engine/_kernel.pyx, compiled by
  oracle/build_ref.py) is driven with OUR flattened tables and seeded rays; its
  complete event logs and tallies are the expected outputs (trace_*.npz,
  tallies_lsc_1e6.npz).
* The reference's pure-Python leaf modules that import without third-party
  packages (data spectra, Distribution, Fresnel/phase helpers, Transformable,
  Sphere, Cylinder, FresnelSurfaceDelegate, engine/recorder.py) give known
  answers for spectra.npz, optics.npz, geometry.npz, transforms.npz, phase.npz,
  surface.npz and recorder_ids.json.  The package __init__ is bypassed with a
  namespace stub (it would pull anytree/trimesh/meshcat, absent here; no
  stand-ins for those are written).
"""
import importlib
import os
import sys
import types

sys.dont_write_bytecode = True   # the reference tree is read-only to this project: importing from it must not leave __pycache__ there

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF = "/root/reference/pvtrace"


def ref_module(name):
    if "pvtrace" not in sys.modules:
        pkg = types.ModuleType("pvtrace")
        pkg.__path__ = [REF]
        sys.modules["pvtrace"] = pkg
    return importlib.import_module(name)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def make_spectra():
    lum = ref_module("pvtrace.data.lumogen_f_red_305")
    flu = ref_module("pvtrace.data.fluro_red")
    dist = ref_module("pvtrace.material.distribution")
    x = np.arange(400, 800)
    xw = np.linspace(300.0, 900.0, 241)
    ems = lum.emission(x)
    d = dist.Distribution(x, ems)
    p = np.linspace(0.0, 1.0, 101)
    xs = np.linspace(400.0, 799.0, 57)
    save(
        "spectra.npz",
        x=x, lumogen_abs=lum.absorption(x), lumogen_ems=ems,
        xw=xw, lumogen_abs_w=lum.absorption(xw), lumogen_ems_w=lum.emission(xw),
        fluro_abs_w=flu.absorption(xw), fluro_ems_w=flu.emission(xw),
        lumogen_ems_cdf=np.asarray(d._cdf),
        sample_p=p, sample_x=np.asarray(d.sample(p)),
        lookup_x=xs, lookup_p=np.asarray(d.lookup(xs)), call_y=np.asarray(d(xs)),
    )


def make_optics():
    mu = ref_module("pvtrace.material.utils")
    rng = np.random.default_rng(11)
    angles = np.concatenate(([0.0], np.sort(rng.uniform(0, np.pi / 2, 400)), [np.pi / 2 - 1e-9]))
    pairs = np.array([(1.0, 1.5), (1.5, 1.0), (1.0, 1.33), (1.49, 1.7), (2.4, 1.0), (1.0, 1.0)])
    refl = np.array([[mu.fresnel_reflectivity(a, n1, n2) for a in angles] for n1, n2 in pairs])
    d = rng.normal(size=(300, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    n = rng.normal(size=(300, 3)); n /= np.linalg.norm(n, axis=1)[:, None]
    spec = np.array([mu.specular_reflection(a, b) for a, b in zip(d, n)])
    # refraction needs the normal flipped along the ray and no TIR
    nf = np.where((np.einsum("ij,ij->i", d, n) < 0)[:, None], -n, n)
    refr_up = np.array([mu.fresnel_refraction(a, b, 1.0, 1.5) for a, b in zip(d, nf)])
    cosang = np.einsum("ij,ij->i", d, nf)
    ok = np.arccos(np.clip(cosang, -1, 1)) < np.arcsin(1.0 / 1.5) - 1e-6
    refr_down = np.array([mu.fresnel_refraction(a, b, 1.5, 1.0) for a, b in zip(d[ok], nf[ok])])
    save("optics.npz", angles=angles, pairs=pairs, reflectivity=refl, d=d, n=n, nf=nf,
         specular=spec, refract_up=refr_up, down_mask=ok, refract_down=refr_down)


def make_geometry():
    sph = ref_module("pvtrace.geometry.sphere")
    cyl = ref_module("pvtrace.geometry.cylinder")
    rng = np.random.default_rng(3)
    n = 400
    o = rng.uniform(-3, 3, size=(n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    sphere = sph.Sphere(1.7)
    cylinder = cyl.Cylinder(2.5, 0.9)

    def hits(geometry):
        out = np.full((n, 4, 3), np.nan)
        cnt = np.zeros(n, dtype=np.int32)
        for i in range(n):
            pts = geometry.intersections(tuple(o[i]), tuple(d[i]))
            cnt[i] = len(pts)
            for k, p in enumerate(pts):
                out[i, k] = p
        return cnt, out

    sc, sp = hits(sphere)
    cc, cp = hits(cylinder)
    sn = np.array([sphere.normal(tuple(p)) for p in sp[sc > 0, 0]])
    cn = np.array([cylinder.normal(tuple(p)) for p in cp[cc > 0, 0]])
    save("geometry.npz", origin=o, direction=d, sphere_radius=1.7, cyl_length=2.5, cyl_radius=0.9,
         sphere_count=sc, sphere_points=sp, sphere_normals=sn,
         cyl_count=cc, cyl_points=cp, cyl_normals=cn)


def make_transforms():
    tr = ref_module("pvtrace.geometry.transformable")
    # nested_cylinders: A then child B; LSC light; kitchen-sink slab
    a = tr.Transformable(); a.translate((0, 0, 2)); a.rotate(np.pi * 0.2, (0, 1, 0))
    b = tr.Transformable(); b.rotate(np.pi / 2, (1, 0, 0))
    light = tr.Transformable(); light.location = (0.0, 0.0, 5.0); light.rotate(np.radians(180), (1, 0, 0))
    slab = tr.Transformable(); slab.translate((0.5, -0.3, 1.0)); slab.rotate(0.35, (1.0, 0.4, 0.2))
    located = tr.Transformable(location=(1.0, 2.0, 3.0)); located.rotate(1.1, (0.0, 1.0, 0.3))
    located.translate((0.5, 0.5, -2.0))
    save("transforms.npz", A=a.pose, B=b.pose, B_in_world=np.dot(a.pose, b.pose), light=light.pose,
         slab=slab.pose, located=located.pose)


def table_dump(compiled):
    return {f"tab_{k}": np.asarray(v) for k, v in compiled.tables().items()}


def make_traces():
    from oracle import oracle as O
    from pvtrace_amd.engine import compile_scene
    from pvtrace_amd.engine.emit import emit_bundle
    from tests import scenes

    for name, mk in scenes.REFERENCE_SCENES.items():
        scene = mk()
        compiled = compile_scene(scene)
        n = 256
        pos, dirs, wl, _ = emit_bundle(scene, n, seed=2024)
        params = dict(seed=99, maxsteps=60 if name == "trapped_light" else 1000, max_events=48,
                      emit_method={"bench_slab": 1, "kitchen_sink": 2}.get(name, 0), record_every=1)
        ref = O.reference_trace_bundle(compiled, pos, dirs, wl, params["seed"], params["maxsteps"],
                                       params["max_events"], params["emit_method"], 1,
                                       params["record_every"])
        arrays = {f"ref_{k}": np.asarray(v) for k, v in ref.items()}
        arrays.update(table_dump(compiled))
        arrays.update(in_pos=pos, in_dir=dirs, in_wl=wl,
                      **{f"par_{k}": np.int64(v) for k, v in params.items()})
        save(f"trace_{name}.npz", **arrays)


def make_lsc_tallies():
    """10^6 photons through the REFERENCE kernel on the headline scene (tallies only)."""
    from oracle import oracle as O
    from pvtrace_amd.engine import compile_scene
    from pvtrace_amd.engine.emit import emit_bundle
    from tests import scenes

    scene = scenes.lsc_equivalent()
    compiled = compile_scene(scene)
    n = 1_000_000
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=777)
    ref = O.reference_trace_bundle(compiled, pos, dirs, wl, 1, 1000, 128, 0, os.cpu_count(), 0)
    checksum = np.array([pos.sum(), dirs.sum(), wl.sum(), np.abs(dirs).sum()])
    save("tallies_lsc_1e6.npz", n=np.int64(n), emit_seed=np.int64(777), seed=np.int64(1),
         input_checksum=checksum, recorder_names=np.array(compiled.recorder_names),
         rec_distinct=ref["rec_distinct"], rec_crossings=ref["rec_crossings"],
         rec_sums=ref["rec_sums"], rec_bins=ref["rec_bins"], **table_dump(compiled))


# Reference-kernel tallies at 10^6 photons for the other configs the reference kernel can run
# (VERDICT r2 #1): BASELINE configs[3] nested_cylinders (examples/nested_cylinders.py:21-64), configs[0]
# hello_world and the reference's own benchmark slab (benchmarks/benchmark_engine.py:26-55), same recipe
# as the headline file.  hello_world has no recorders in the example; whole-surface recorders are attached
# here so that there is something to tally.
TALLY_SCENES = {
    "nested_cylinders": dict(scene="nested_cylinders", emit_seed=4104, seed=5, emit_method=0),
    "hello_world": dict(scene="hello_world_recorded", emit_seed=4105, seed=6, emit_method=0),
    "bench_slab": dict(scene="bench_slab_recorded", emit_seed=4106, seed=7, emit_method=1),
    # the scene-size family: 6 x 6 tiles of the headline slab, 37 nodes -- on the GPU this is the
    # node-grid path, in the reference the loop over every node (_kernel.pyx:666-680)
    "tiles6": dict(scene="tiles6", emit_seed=4107, seed=8, emit_method=0),
}


def make_config_tallies(only=()):
    from oracle import oracle as O
    from pvtrace_amd.engine import compile_scene
    from pvtrace_amd.engine.emit import emit_bundle
    from tests import scenes

    for name, spec in TALLY_SCENES.items():
        if only and name not in only:
            continue
        scene = scenes.TALLY_SCENES[spec["scene"]]()
        compiled = compile_scene(scene)
        n = 1_000_000
        pos, dirs, wl, _ = emit_bundle(scene, n, seed=spec["emit_seed"])
        ref = O.reference_trace_bundle(compiled, pos, dirs, wl, spec["seed"], 1000, 128, spec["emit_method"],
                                       os.cpu_count(), 0)
        checksum = np.array([pos.sum(), dirs.sum(), wl.sum(), np.abs(dirs).sum()])
        save(f"tallies_{name}_1e6.npz", n=np.int64(n), emit_seed=np.int64(spec["emit_seed"]),
             seed=np.int64(spec["seed"]), emit_method=np.int64(spec["emit_method"]),
             input_checksum=checksum, recorder_names=np.array(compiled.recorder_names),
             rec_distinct=ref["rec_distinct"], rec_crossings=ref["rec_crossings"],
             rec_sums=ref["rec_sums"], rec_bins=ref["rec_bins"], **table_dump(compiled))


def make_hist_spectra():
    """Histogram-sampled Distribution (hist=True) known answers from the reference class."""
    dist = ref_module("pvtrace.material.distribution")
    rng = np.random.default_rng(8)
    x = np.array([400.0, 410.0, 425.0, 430.0, 455.0, 500.0, 520.0, 600.0, 610.0, 700.0])
    y = np.array([0.0, 0.0, 1.0, 3.0, 2.0, 0.0, 0.0, 4.0, 1.0, 0.0])     # zero runs -> CDF plateaus
    d = dist.Distribution(x, y, hist=True)
    q = np.concatenate((x, rng.uniform(400.0, 700.0, 200)))
    p = np.concatenate(([0.0, 1.0], np.asarray(d._cdf), rng.random(200)))
    save("spectra_hist.npz", x=x, y=y, cdf=np.asarray(d._cdf), query_x=q, value=np.asarray(d(q)),
         lookup=np.asarray(d.lookup(q)), query_p=p, sample=np.asarray(d.sample(p)))


def make_phase():
    """Directions of the reference's phase functions (material/utils.py:104-186) for given draws: numpy's global
    generator is replaced, for the duration of a call, by one that hands out the first draws of our per-ray stream
    (oracle.uniforms(seed, .)), so the same draws can be given to the restatement."""
    from oracle import oracle as O

    mu = ref_module("pvtrace.material.utils")
    seeds = np.arange(1, 201, dtype=np.int64)
    cases = [("isotropic", 0, 0.0, lambda: mu.isotropic()), ("hg", 1, 0.6, lambda: mu.henyey_greenstein(0.6)),
             ("hg_back", 1, -0.35, lambda: mu.henyey_greenstein(-0.35)), ("cone", 2, 0.4, lambda: mu.cone(0.4)),
             ("lambertian", 3, 0.0, lambda: mu.lambertian())]
    out = {"seeds": seeds, "draws": np.array([O.uniforms(int(sd), 2) for sd in seeds])}
    real = np.random.uniform
    for name, tag, param, fn in cases:
        dirs = []
        for sd in seeds:
            feed = list(O.uniforms(int(sd), 2))

            def fake(low=0.0, high=1.0, size=None):
                assert (low, high) == (0.0, 1.0) or (low, high) == (0, 1)
                if size is None:
                    return feed.pop(0)
                return np.array([feed.pop(0) for _ in range(int(size))])

            np.random.uniform = fake
            try:
                dirs.append(np.asarray(fn(), dtype=float))
            finally:
                np.random.uniform = real
        out[f"{name}_tag"] = np.int64(tag)
        out[f"{name}_param"] = np.float64(param)
        out[f"{name}_dir"] = np.array(dirs)
    save("phase.npz", **out)


def make_surface():
    """The reference's FresnelSurfaceDelegate (material/surface.py:102-177) asked about rays meeting a reference Sphere
    and a reference Cylinder: reflectivity, reflected and transmitted direction.  Ray / node arguments are plain
    namespaces with the attributes the delegate reads (ray.position, ray.direction, node.geometry.material
    .refractive_index); the geometry IS the reference's."""
    surf = ref_module("pvtrace.material.surface")
    sph = ref_module("pvtrace.geometry.sphere")
    cyl = ref_module("pvtrace.geometry.cylinder")
    delegate = surf.FresnelSurfaceDelegate()
    rng = np.random.default_rng(21)

    def node(n):
        return types.SimpleNamespace(geometry=types.SimpleNamespace(material=types.SimpleNamespace(refractive_index=n)))

    shapes = {"sphere": (sph.Sphere(1.7), 1, [1.7]), "cyl": (cyl.Cylinder(2.5, 0.9), 2, [2.5, 0.9])}
    out = {}
    for name, (geometry, gtype, params) in shapes.items():
        n = 300
        pts = []
        while len(pts) < n:   # points ON the surface: first intersection of random rays
            o = rng.uniform(-3, 3, 3); d = rng.normal(size=3); d /= np.linalg.norm(d)
            hit = geometry.intersections(tuple(o), tuple(d))
            if hit:
                pts.append(hit[0])
        pts = np.array(pts, dtype=float)
        dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        pairs = np.array([(1.0, 1.5), (1.5, 1.0), (1.33, 1.0), (1.0, 1.0), (1.49, 1.7)])[rng.integers(0, 5, n)]
        refl_r, refl_d, trans_d, normals = [], [], [], []
        for p_, d_, (n1, n2) in zip(pts, dirs, pairs):
            ray = types.SimpleNamespace(position=tuple(p_), direction=tuple(d_))
            args = (None, ray, geometry, node(n1), node(n2))
            r = delegate.reflectivity(*args)
            refl_r.append(r)
            refl_d.append(delegate.reflected_direction(*args))
            with np.errstate(invalid="ignore"):
                trans_d.append(delegate.transmitted_direction(*args) if r < 1.0 else (np.nan,) * 3)
            normals.append(geometry.normal(tuple(p_)))
        out.update({f"{name}_type": np.int64(gtype), f"{name}_params": np.array(params), f"{name}_points": pts,
                    f"{name}_directions": dirs, f"{name}_indices": pairs, f"{name}_normals": np.array(normals, dtype=float),
                    f"{name}_reflectivity": np.array(refl_r), f"{name}_reflected": np.array(refl_d, dtype=float),
                    f"{name}_transmitted": np.array(trans_d, dtype=float)})
    save("surface.npz", **out)


def make_py_tracer():
    """Histories of the REFERENCE's per-ray Python tracer (algorithm/photon_tracer.py:276-328 `follow`, with its
    `find_container`, `next_hit`, `step_forward`) -- the semantic ground truth of SURVEY §8(a)'s last row.  Its scene graph
    (scene/node.py, scene/scene.py) imports anytree, which is not here, and no stand-in for anytree is written: the
    PRODUCT's Node / Scene are put in its place (`sys.modules`), i.e. the reference's tracer walks the product's tree and
    calls the product's `Scene.intersections` / `Node.point_to_node` -- while geometry (Sphere, Cylinder), Material,
    components, surface delegate, Distribution and Ray are all the reference's own.  Every ray is traced under its own
    numpy seed; `oracle/py_tracer.py` must reproduce every history from the same seeds (tests/test_py_tracer.py)."""
    import pvtrace_amd.scene as prod_scene
    from tests import scenes

    ref_module("pvtrace.data.lumogen_f_red_305")
    for sub in ("scene", "light", "material", "geometry", "algorithm", "common"):
        if f"pvtrace.{sub}" not in sys.modules:
            pkg = types.ModuleType(f"pvtrace.{sub}")
            pkg.__path__ = [os.path.join(REF, sub)]
            sys.modules[f"pvtrace.{sub}"] = pkg
    sys.modules["pvtrace.scene.node"] = prod_scene
    sys.modules["pvtrace.scene.scene"] = prod_scene
    tracer = ref_module("pvtrace.algorithm.photon_tracer")
    ray_cls = ref_module("pvtrace.light.ray").Ray
    comp = ref_module("pvtrace.material.component")
    classes = types.SimpleNamespace(
        Sphere=ref_module("pvtrace.geometry.sphere").Sphere, Cylinder=ref_module("pvtrace.geometry.cylinder").Cylinder,
        Material=ref_module("pvtrace.material.material").Material, Luminophore=comp.Luminophore, Absorber=comp.Absorber,
        Scatterer=comp.Scatterer, lumogen=ref_module("pvtrace.data.lumogen_f_red_305"))
    scene = scenes.py_tracer_pin_scene(classes)
    dirs, wls, seeds = scenes.py_tracer_pin_rays()
    counts, kinds, pos, direc, wl = [], [], [], [], []
    for d, w, sd in zip(dirs, wls, seeds):
        np.random.seed(int(sd))
        hist = tracer.follow(scene, ray_cls(position=(0.0, 0.0, 0.0), direction=tuple(d), wavelength=float(w)))
        counts.append(len(hist))
        for ray, event in hist:
            kinds.append(event.value); pos.append(ray.position); direc.append(ray.direction); wl.append(ray.wavelength)
    save("py_tracer.npz", directions=dirs, wavelengths=wls, seeds=seeds, counts=np.array(counts), kind=np.array(kinds),
         position=np.array(pos, dtype=float), direction=np.array(direc, dtype=float), wavelength=np.array(wl, dtype=float))
    print("   events:", np.bincount(np.array(kinds), minlength=10).tolist())


def make_emit():
    """The reference's host emitter, `engine/emit.py:92-134 emit_bundle`, on five posed lights with every built-in
    delegate (the reference's own Light, mask and phase-function classes on the product's Nodes; its scene graph needs
    anytree, see make_py_tracer), under numpy seeds: the product's `emit_bundle(scene, n, seed=None)` draws from the same
    global generator in the same order and must return the same arrays, bit for bit (tests/test_scene_api.py)."""
    import pvtrace_amd.scene as prod_scene
    from tests import scenes

    ref_module("pvtrace.data.lumogen_f_red_305")
    for sub in ("scene", "light", "material", "geometry", "algorithm", "common", "engine"):
        if f"pvtrace.{sub}" not in sys.modules:
            pkg = types.ModuleType(f"pvtrace.{sub}")
            pkg.__path__ = [os.path.join(REF, sub)]
            sys.modules[f"pvtrace.{sub}"] = pkg
    sys.modules["pvtrace.scene.node"] = prod_scene
    sys.modules["pvtrace.scene.scene"] = prod_scene
    emit = ref_module("pvtrace.engine.emit")
    full = scenes.emit_pin_scene(ref_module("pvtrace.light.light"), ref_module("pvtrace.material.utils"),
                                 ref_module("pvtrace.material.distribution").Distribution)
    # (the product's Scene.light_nodes looks for the product's Light class; the reference's emitter needs two attributes)
    scene = types.SimpleNamespace(root=full.root, light_nodes=[n for n in full.root.levelorder() if getattr(n, "light", None) is not None])
    out = {}
    for n, seed in ((1, 11), (7, 12), (1003, 13)):
        np.random.seed(seed)
        pos, direc, wl, sources = emit.emit_bundle(scene, n)
        out.update({f"n{n}_seed": np.int64(seed), f"n{n}_position": pos, f"n{n}_direction": direc, f"n{n}_wavelength": wl,
                    f"n{n}_sources": np.array(sources)})
    save("emit.npz", **out)


LSC_DELEGATE_CONFIGS = [   # (back-surface mirror, solar-cell edges)
    (False, ()), (True, ()), (True, ("left", "right", "near", "far")), (False, ("left", "far")),
]


def lsc_delegate_rays(seed=31, per_face=14):
    """Rays meeting the six faces of a 5 x 5 x 1 box from inside (n 1.5 -> 1.0) and from outside (1.0 -> 1.5):
    (positions on the faces, directions, n1, n2).  Shared by the generator and tests/test_coating_known_rays.py."""
    rng = np.random.default_rng(seed)
    half = np.array([2.5, 2.5, 0.5])
    pos, direc, n1, n2 = [], [], [], []
    for axis in range(3):
        for sign in (-1.0, 1.0):
            for k in range(per_face):
                p = rng.uniform(-0.95, 0.95, 3) * half
                p[axis] = sign * half[axis]
                d = rng.normal(size=3)
                d /= np.linalg.norm(d)
                outward = d[axis] * sign > 0
                if (k % 2 == 0) != outward:   # even k: leaving the box, odd k: arriving from outside
                    d[axis] = -d[axis]
                pos.append(p); direc.append(d)
                n1.append(1.5 if k % 2 == 0 else 1.0); n2.append(1.0 if k % 2 == 0 else 1.5)
    return np.array(pos), np.array(direc), np.array(n1), np.array(n2)


def make_lsc_delegates():
    """The reference's LSC surface delegates (`device/lsc.py:22-86`: OptionalMirrorAndSolarCell, AirGapMirror) asked
    about rays on every face of the slab, in four configurations.  `device/lsc.py` imports the reference's trimesh-backed
    Box, its anytree-based scene graph and its meshcat renderer: the product's geometry / scene modules are put in the
    place of the first two (as in make_py_tracer) and an EMPTY object in the place of the renderer module (a visualiser,
    nothing of the path) -- no third-party package is imitated.  The box the delegates ask for normals is the product's
    (itself held to the reference's known answers, tests/test_golden_units.py)."""
    import pvtrace_amd.geometry as prod_geometry
    import pvtrace_amd.scene as prod_scene

    ref_module("pvtrace.data.lumogen_f_red_305")
    for sub in ("scene", "light", "material", "geometry", "algorithm", "common", "device"):
        if f"pvtrace.{sub}" not in sys.modules:
            pkg = types.ModuleType(f"pvtrace.{sub}")
            pkg.__path__ = [os.path.join(REF, sub)]
            sys.modules[f"pvtrace.{sub}"] = pkg
    sys.modules["pvtrace.scene.node"] = prod_scene
    sys.modules["pvtrace.scene.scene"] = prod_scene
    sys.modules["pvtrace.geometry.box"] = prod_geometry
    sys.modules["pvtrace.scene.renderer"] = types.SimpleNamespace(MeshcatRenderer=None)
    lsc = ref_module("pvtrace.device.lsc")
    box = prod_geometry.Box((5.0, 5.0, 1.0))
    pos, direc, n1, n2 = lsc_delegate_rays()

    def node(n):
        return types.SimpleNamespace(geometry=types.SimpleNamespace(material=types.SimpleNamespace(refractive_index=n)))

    out = {"positions": pos, "directions": direc, "n1": n1, "n2": n2}
    for c, (mirror, cells) in enumerate(LSC_DELEGATE_CONFIGS):
        owner = types.SimpleNamespace(_solar_cell_surfaces=set(cells), _back_surface_mirror_info={"want_back_surface_mirror": mirror},
                                      _air_gap_mirror_info={"want_air_gap_mirror": True, "lambertian": False})
        delegate = lsc.OptionalMirrorAndSolarCell(owner)
        refl, trans, spec = [], [], []
        for p_, d_, a, b in zip(pos, direc, n1, n2):
            ray = types.SimpleNamespace(position=tuple(p_), direction=tuple(d_))
            args = (None, ray, box, node(a), node(b))
            r = delegate.reflectivity(*args)
            refl.append(r)
            spec.append(delegate.reflected_direction(*args))
            with np.errstate(invalid="ignore"):
                trans.append(delegate.transmitted_direction(*args) if r < 1.0 else (np.nan,) * 3)
        out.update({f"cfg{c}_reflectivity": np.array(refl, dtype=float), f"cfg{c}_reflected": np.array(spec, dtype=float),
                    f"cfg{c}_transmitted": np.array(trans, dtype=float)})
        if c == 0:
            gap = lsc.AirGapMirror(owner)
            out["airgap_reflectivity"] = np.array([gap.reflectivity(None, types.SimpleNamespace(position=tuple(p_), direction=tuple(d_)),
                                                                    box, node(a), node(b)) for p_, d_, a, b in zip(pos, direc, n1, n2)])
    save("lsc_delegates.npz", **out)


def make_lsc_scenes():
    """The scenes the reference's `LSC` class builds (`device/lsc.py:95-219`: `_make_scene` through its public `add_*`
    methods), described node by node -- names, box sizes, refractive indices, poses, components with their spectra, lights
    with their delegates.  BASELINE configs[1] is `LSC((5, 5, 1))`: the product's builder must make the same scene
    (tests/test_scene_api.py).  Module substitutions as in make_lsc_delegates."""
    import pvtrace_amd.geometry as prod_geometry
    import pvtrace_amd.scene as prod_scene
    from tests import scenes
    from tests.util import describe_lsc_scene

    ref_module("pvtrace.data.lumogen_f_red_305")
    for sub in ("scene", "light", "material", "geometry", "algorithm", "common", "device"):
        if f"pvtrace.{sub}" not in sys.modules:
            pkg = types.ModuleType(f"pvtrace.{sub}")
            pkg.__path__ = [os.path.join(REF, sub)]
            sys.modules[f"pvtrace.{sub}"] = pkg
    sys.modules["pvtrace.scene.node"] = prod_scene
    sys.modules["pvtrace.scene.scene"] = prod_scene
    sys.modules["pvtrace.geometry.box"] = prod_geometry
    sys.modules.setdefault("pvtrace.scene.renderer", types.SimpleNamespace(MeshcatRenderer=None))
    lsc = ref_module("pvtrace.device.lsc")
    light = ref_module("pvtrace.light.light")
    utils = ref_module("pvtrace.material.utils")
    cases = scenes.lsc_builder_cases(lsc.LSC, utils.cone, light.rectangular_mask, ref_module("pvtrace.data.lumogen_f_red_305"))
    out = {}
    for name, device in cases.items():
        device._make_scene()
        for key, value in describe_lsc_scene(device._scene).items():
            out[f"{name}/{key}"] = value
    save("lsc_scenes.npz", **out)


LSC_TRACER_RAYS = {"default": 3000, "cells": 3000, "custom": 3000}


def make_lsc_tracer():
    """the reference's per-ray Python tracer (`algorithm/photon_tracer.py follow`) run on the scenes the
    reference's `LSC` class builds, CALLING the reference's LSC surface delegates (solar cells, back-surface mirror,
    air-gap mirror, specular and lambertian) at every hit -- the semantic ground truth for what the product lowers to
    coating tables.  Rays come from the reference's `emit_bundle` under a numpy seed.  Kept: per-ray event counts by
    kind, the last event's kind, and the position the reference's `LSC.simulate` would store as the exit ray
    (`device/lsc.py:349-359`: the last ray for ABSORB / KILL, the one before the last for EXIT).  Compared
    statistically (Welch, 5 sigma) with the C referee and with the GPU engine on the PRODUCT's `LSC` of the same
    configuration.  Module substitutions as in make_lsc_delegates."""
    import pvtrace_amd.geometry as prod_geometry
    import pvtrace_amd.scene as prod_scene
    from tests import scenes

    ref_module("pvtrace.data.lumogen_f_red_305")
    for sub in ("scene", "light", "material", "geometry", "algorithm", "common", "device", "engine"):
        if f"pvtrace.{sub}" not in sys.modules:
            pkg = types.ModuleType(f"pvtrace.{sub}")
            pkg.__path__ = [os.path.join(REF, sub)]
            sys.modules[f"pvtrace.{sub}"] = pkg
    sys.modules["pvtrace.scene.node"] = prod_scene
    sys.modules["pvtrace.scene.scene"] = prod_scene
    sys.modules["pvtrace.geometry.box"] = prod_geometry
    sys.modules.setdefault("pvtrace.scene.renderer", types.SimpleNamespace(MeshcatRenderer=None))
    lsc = ref_module("pvtrace.device.lsc")
    light = ref_module("pvtrace.light.light")
    utils = ref_module("pvtrace.material.utils")
    emit = ref_module("pvtrace.engine.emit")
    tracer = ref_module("pvtrace.algorithm.photon_tracer")
    ray_cls = ref_module("pvtrace.light.ray").Ray
    event_cls = ref_module("pvtrace.light.event").Event
    cases = scenes.lsc_tracer_cases(lsc.LSC, utils.cone, light.rectangular_mask, ref_module("pvtrace.data.lumogen_f_red_305"))
    out = {}
    for c, (name, device) in enumerate(cases.items()):
        device._make_scene()
        full = device._scene
        view = types.SimpleNamespace(root=full.root, light_nodes=[n for n in full.root.levelorder() if getattr(n, "light", None) is not None])
        n = LSC_TRACER_RAYS[name]
        np.random.seed(700 + c)
        pos, direc, wl, _ = emit.emit_bundle(view, n)
        counts = np.zeros((n, 10), dtype=np.uint16)
        last = np.zeros(n, dtype=np.uint8)
        where = np.zeros((n, 3))
        for j in range(n):
            hist = tracer.follow(full, ray_cls(position=tuple(pos[j]), direction=tuple(direc[j]), wavelength=float(wl[j])))
            for _, event in hist:
                counts[j, event.value] += 1
            last[j] = hist[-1][1].value
            where[j] = hist[-2][0].position if hist[-1][1] == event_cls.EXIT else hist[-1][0].position
        out.update({f"{name}/counts": counts, f"{name}/last": last, f"{name}/where": where})
        print("  ", name, "mean events per ray by kind value:", np.round(counts.mean(axis=0), 3).tolist())
    save("lsc_tracer.npz", **out)


CFG5_TRACER_RAYS = 20000


def make_cfg5_tracer():
    """SURVEY §8(c)(5) / §8(d) cfg5, as they are written: BASELINE configs[4] traced by the reference's OWN per-ray Python
    tracer (`algorithm/photon_tracer.py:276-328` `follow`) calling the Coatings notebook's OWN delegate.  The class
    `PartialTopSurfaceMirror` is read from `examples/006 Coatings.ipynb` where it lies (`json.load`, cell 3) and `exec`'d
    against the reference's `FresnelSurfaceDelegate` -- nothing of it is stored here; the scene is the notebook's cell 5
    (world Box 15^3, slab Box (10,10,1) of index 1.5 with that delegate, `Light(position=partial(rectangular_mask, 5, 5))`
    at (0,0,2) turned by 180 degrees about x) plus the `Scatterer(1.0)` of benchmarks/configs.py:cfg5_coated_slab, built
    from the reference's Material / Surface / Scatterer / Light.  Module substitutions as in make_lsc_tracer: the
    reference's scene graph needs anytree and its Box needs trimesh, neither is here and neither is imitated -- the
    PRODUCT's Node / Scene / Box stand in their place, so the tree traversal and the ray-box arithmetic under the
    reference's tracer are the product's (each held to the reference's known answers elsewhere: tests/test_golden_units.py,
    tests/test_intersection.py); the tracer, the delegate, the material, the scatterer and the emission are the reference's.
    Kept per ray (numbers only): event counts by kind, the last event, and the position the reference's `LSC.simulate`
    would store as the exit ray.  Rays: the reference's `emit_bundle` under numpy seed 905."""
    import functools
    import json

    import pvtrace_amd.geometry as prod_geometry
    import pvtrace_amd.scene as prod_scene

    ref_module("pvtrace.data.lumogen_f_red_305")
    for sub in ("scene", "light", "material", "geometry", "algorithm", "common", "device", "engine"):
        if f"pvtrace.{sub}" not in sys.modules:
            pkg = types.ModuleType(f"pvtrace.{sub}")
            pkg.__path__ = [os.path.join(REF, sub)]
            sys.modules[f"pvtrace.{sub}"] = pkg
    sys.modules["pvtrace.scene.node"] = prod_scene
    sys.modules["pvtrace.scene.scene"] = prod_scene
    sys.modules["pvtrace.geometry.box"] = prod_geometry
    surface = ref_module("pvtrace.material.surface")
    material = ref_module("pvtrace.material.material")
    component = ref_module("pvtrace.material.component")
    light = ref_module("pvtrace.light.light")
    emit = ref_module("pvtrace.engine.emit")
    tracer = ref_module("pvtrace.algorithm.photon_tracer")
    ray_cls = ref_module("pvtrace.light.ray").Ray
    event_cls = ref_module("pvtrace.light.event").Event

    notebook = json.load(open(os.path.join(os.path.dirname(REF), "examples", "006 Coatings.ipynb")))
    cells = ["".join(c["source"]) for c in notebook["cells"] if c["cell_type"] == "code"]
    (source,) = [c for c in cells if c.lstrip().startswith("class PartialTopSurfaceMirror")]
    namespace = {"np": np, "FresnelSurfaceDelegate": surface.FresnelSurfaceDelegate}
    exec(compile(source, "006 Coatings.ipynb cell 3", "exec"), namespace)   # noqa: S102 -- the notebook's class, where it lies
    mirror_cls = namespace["PartialTopSurfaceMirror"]

    world = prod_scene.Node(name="world (air)", geometry=prod_geometry.Box((15.0, 15.0, 15.0), material=material.Material(refractive_index=1.0)))
    prod_scene.Node(name="box (glass)", parent=world, geometry=prod_geometry.Box(
        (10.0, 10.0, 1.0), material=material.Material(refractive_index=1.5, surface=surface.Surface(delegate=mirror_cls()),
                                                      components=[component.Scatterer(1.0, name="Scatterer")])))
    lamp = prod_scene.Node(name="Light", parent=world,
                           light=light.Light(position=functools.partial(light.rectangular_mask, 5, 5), name="Light"))
    lamp.location = (0, 0, 2)
    lamp.rotate(np.radians(180), (1, 0, 0))
    full = prod_scene.Scene(world)
    view = types.SimpleNamespace(root=full.root, light_nodes=[n for n in full.root.levelorder() if getattr(n, "light", None) is not None])
    n = CFG5_TRACER_RAYS
    np.random.seed(905)
    pos, direc, wl, _ = emit.emit_bundle(view, n)
    counts = np.zeros((n, 10), dtype=np.uint16)
    last = np.zeros(n, dtype=np.uint8)
    where = np.zeros((n, 3))
    for j in range(n):
        hist = tracer.follow(full, ray_cls(position=tuple(pos[j]), direction=tuple(direc[j]), wavelength=float(wl[j])))
        for _, event in hist:
            counts[j, event.value] += 1
        last[j] = hist[-1][1].value
        where[j] = hist[-2][0].position if hist[-1][1] == event_cls.EXIT else hist[-1][0].position
    print("   cfg5: mean events per ray by kind value:", np.round(counts.mean(axis=0), 4).tolist())
    save("cfg5_tracer.npz", **{"cfg5/counts": counts, "cfg5/last": last, "cfg5/where": where,
                               "cfg5/first_positions": pos[:64], "cfg5/first_directions": direc[:64]})


def make_object_methods():
    """The per-interaction methods of the reference's host objects -- `Material.penetration_depth / is_absorbed / component`
    (material/material.py:22-63), `Scatterer / Absorber / Luminophore .is_radiative / nonradiative_absorb / emit`
    (material/component.py:168-196, :236-239, :381-440), `Surface.is_reflected / reflect / transmit` (material/surface.py:
    224-272) -- called in a fixed order under numpy seeds by `tests/scenes.py::object_method_script`; the product's classes
    run the same script and must return the same numbers, bit for bit (tests/test_golden_units.py).  All of these modules
    import without third-party packages but for `light/ray.py`, which names the anytree-based Node: the product's stands there."""
    import pvtrace_amd.scene as prod_scene
    from tests import scenes

    ref_module("pvtrace.data.lumogen_f_red_305")
    for sub in ("scene", "light", "material", "geometry", "common", "engine"):
        if f"pvtrace.{sub}" not in sys.modules:
            pkg = types.ModuleType(f"pvtrace.{sub}")
            pkg.__path__ = [os.path.join(REF, sub)]
            sys.modules[f"pvtrace.{sub}"] = pkg
    sys.modules.setdefault("pvtrace.scene.node", prod_scene)   # (light/ray.py names Node for `representation`, unused here)
    c = types.SimpleNamespace(**{k: v for k, v in reference_classes_for_scenes().items()
                                 if k in ("Material", "Absorber", "Scatterer", "Reactor", "Luminophore", "Surface",
                                          "NullSurfaceDelegate", "Sphere", "cone")})
    c.henyey_greenstein = ref_module("pvtrace.material.utils").henyey_greenstein
    c.Ray = ref_module("pvtrace.light.ray").Ray
    c.lumogen = ref_module("pvtrace.data.lumogen_f_red_305")
    save("object_methods.npz", **scenes.object_method_script(c))


def reference_classes_for_scenes():
    """{name in tests/scenes.py: the reference's object of that name} -- geometry (but Box), materials, components, surfaces,
    lights and masks, phase functions, recorders, spectra data -- for building the twins of the test scenes."""
    comp, surf, light, utils = (ref_module("pvtrace.material.component"), ref_module("pvtrace.material.surface"),
                                ref_module("pvtrace.light.light"), ref_module("pvtrace.material.utils"))
    rec = ref_module("pvtrace.engine.recorder")
    return dict(
        Absorber=comp.Absorber, Luminophore=comp.Luminophore, Scatterer=comp.Scatterer, Reactor=comp.Reactor,
        Material=ref_module("pvtrace.material.material").Material, Surface=surf.Surface, NullSurfaceDelegate=surf.NullSurfaceDelegate,
        Sphere=ref_module("pvtrace.geometry.sphere").Sphere, Cylinder=ref_module("pvtrace.geometry.cylinder").Cylinder,
        Light=light.Light, rectangular_mask=light.rectangular_mask, CircularMask=light.CircularMask,
        ConstantWavelengthMask=light.ConstantWavelengthMask, CubeMask=light.CubeMask, SpectrumWavelengthMask=light.SpectrumWavelengthMask,
        cone=utils.cone, isotropic=utils.isotropic, lambertian=utils.lambertian, Cone=utils.Cone, HenyeyGreenstein=utils.HenyeyGreenstein,
        gaussian=utils.gaussian, Distribution=ref_module("pvtrace.material.distribution").Distribution,
        Recorder=rec.Recorder, Histogram=rec.Histogram, Heatmap=rec.Heatmap, lumogen_f_red_305=ref_module("pvtrace.data.lumogen_f_red_305"))


def make_compiled_tables():
    """The flat tables of the REFERENCE's own flattener, `engine/compiler.py:57-331 compile_scene`, for the eight test
    scenes the reference engine can express -- the flattener's pin from outside (it had been hand-restated tables only).
    The scenes are built by tests/scenes.py's own builders with the reference's classes put under the names they use
    (materials, components, surfaces, Sphere, Cylinder, lights, masks, phase functions, recorders); Node / Scene / Box are the
    product's, standing where the reference's anytree- and trimesh-based ones are looked for, and the ONE name the
    reference compiler takes from anytree, `PreOrderIter`, is bound to the product tree's own pre-order walk
    (`Node.preorder`): an adapter to the tree that is there, not an imitation of the package."""
    import contextlib

    import pvtrace_amd.geometry as prod_geometry
    import pvtrace_amd.scene as prod_scene
    from tests import scenes

    ref_module("pvtrace.data.lumogen_f_red_305")
    for sub in ("scene", "light", "material", "geometry", "algorithm", "common", "device", "engine"):
        if f"pvtrace.{sub}" not in sys.modules:
            pkg = types.ModuleType(f"pvtrace.{sub}")
            pkg.__path__ = [os.path.join(REF, sub)]
            sys.modules[f"pvtrace.{sub}"] = pkg
    sys.modules["pvtrace.scene.node"] = prod_scene
    sys.modules["pvtrace.scene.scene"] = prod_scene
    sys.modules["pvtrace.geometry.box"] = prod_geometry
    sys.modules["pvtrace.geometry.mesh"] = prod_geometry
    sys.modules["anytree"] = types.SimpleNamespace(PreOrderIter=lambda root: root.preorder())
    compiler = ref_module("pvtrace.engine.compiler")
    theirs = reference_classes_for_scenes()

    @contextlib.contextmanager
    def reference_names():
        old = {k: getattr(scenes, k) for k in theirs if hasattr(scenes, k)}
        for k in old:
            setattr(scenes, k, theirs[k])
        try:
            yield
        finally:
            for k, v in old.items():
                setattr(scenes, k, v)

    out = {}
    fields = ("geom_type", "geom_params", "local_to_world", "world_to_local", "refractive_index", "surface_type", "comp_start",
              "comp_count", "comp_type", "comp_qy", "comp_tau_rad", "comp_tau_nr", "comp_phase_type", "comp_phase_param",
              "comp_abs_start", "comp_abs_n", "comp_ems_start", "comp_ems_n", "abs_x", "abs_y", "ems_x", "ems_cdf", "rec_node",
              "rec_event", "rec_has_facet", "rec_facet", "rec_atol", "rec_hist_start", "rec_hist_n", "hist_prop_a", "hist_prop_b",
              "hist_na", "hist_nb", "hist_lo_a", "hist_hi_a", "hist_lo_b", "hist_hi_b", "hist_offset")
    for name, build in scenes.REFERENCE_SCENES.items():
        with reference_names():
            scene = build()
        compiled = compiler.compile_scene(scene)
        for f in fields:
            out[f"{name}/{f}"] = np.asarray(getattr(compiled, f))
        out[f"{name}/total_bins"] = np.int64(compiled.total_bins)
        out[f"{name}/root_id"] = np.int64(compiled.root_id)
        out[f"{name}/node_names"] = np.array(compiled.node_names)
        out[f"{name}/component_names"] = np.array(list(compiled.component_names) or [""])
        out[f"{name}/recorder_names"] = np.array([spec if isinstance(spec, str) else spec[0] for spec in getattr(compiled, "recorder_names", [])] or [""])
    save("compiled_tables.npz", **out)


def describe_recorders(recorders, moment_properties=("wavelength", "angle", "duration", "pathlength")):
    """{name: RecorderResult} -> flat {key: array}: counts, mean / std / error of the four properties, histograms."""
    out, names = {}, []
    for name, rec in recorders.items():
        names.append(name)
        out[f"rec/{name}/counts"] = np.array([rec.rays, rec.crossings], dtype=np.int64)
        out[f"rec/{name}/stats"] = np.array([[rec.mean(p), rec.std(p), rec.error(p)] for p in moment_properties], dtype=float)
        for h in range(len(rec.spec.histograms)):
            for k, part in enumerate(rec.histogram(h)):
                out[f"rec/{name}/hist{h}/{k}"] = np.asarray(part)
    out["recorder_names"] = np.array(names or [""])
    return out


def describe_engine_result(result, moment_properties=("wavelength", "angle", "duration", "pathlength")):
    """Everything an `EngineResult` answers (reference api.py:81-194), as a flat {key: array} dict -- the reference's and
    the product's alike (tests/test_scene_api.py imports this function)."""
    out = {"num_rays": np.int64(result.num_rays), "num_recorded": np.int64(result.num_recorded),
           "recorded_indices": np.asarray(result.recorded_indices)}
    names = []
    for name, rec in result.recorders.items():
        names.append(name)
        out[f"rec/{name}/counts"] = np.array([rec.rays, rec.crossings], dtype=np.int64)
        out[f"rec/{name}/stats"] = np.array([[rec.mean(p), rec.std(p), rec.error(p)] for p in moment_properties], dtype=float)
        for h in range(len(rec.spec.histograms)):
            parts = rec.histogram(h)
            for k, part in enumerate(parts):
                out[f"rec/{name}/hist{h}/{k}"] = np.asarray(part)
    out["recorder_names"] = np.array(names or [""])
    counts = result.event_counts()
    out["event_counts"] = np.array([counts.get(type(next(iter(counts)))(v), 0) if counts else 0 for v in range(10)], dtype=np.int64)
    rows, texts, lengths = [], [], []
    for history in result.histories():
        lengths.append(len(history))
        for ray, event, meta in history:
            normal = meta.get("normal", (np.nan, np.nan, np.nan))
            rows.append([event.value, *ray.position, *ray.direction, ray.wavelength, ray.travelled, ray.duration, *normal])
            texts.append("|".join(str(x) for x in (ray.source, meta["hit"], meta["container"], meta["adjacent"], meta["component"],
                                                   sorted(meta))))
    out["history_lengths"] = np.array(lengths, dtype=np.int64)
    out["history_rows"] = np.array(rows, dtype=float).reshape(len(rows), 13)
    out["history_texts"] = np.array(texts or [""])
    return out


def make_engine_result():
    """The reference's whole host pipeline around its kernel -- `compile_scene` (compiler.py), `emit_bundle` (emit.py, under a
    numpy seed), `_kernel.trace_bundle` (its own compiled kernel, oracle/_ref), `EngineResult` / `RecorderResult` (api.py:26-194)
    -- on the kitchen-sink scene built with the reference's classes (substitutions: make_compiled_tables).  Saved: the
    kernel's raw `data` dict with the emitted rays' sources, and everything the reference's result object answers about
    it.  The product's `EngineResult` on the SAME `data` and the product's tables must answer the same
    (tests/test_scene_api.py): recorder counts, moments, histogram edges and counts, event counts, and every history with
    its Ray fields, source names and metadata."""
    import contextlib

    from oracle import oracle as O
    from tests import scenes

    make_compiled_tables()   # (sets the substitutions up; its own output is rewritten identically)
    compiler = ref_module("pvtrace.engine.compiler")
    api = ref_module("pvtrace.engine.api")
    emit = ref_module("pvtrace.engine.emit")
    theirs = reference_classes_for_scenes()
    old = {k: getattr(scenes, k) for k in theirs if hasattr(scenes, k)}
    for k in old:
        setattr(scenes, k, theirs[k])
    try:
        full = scenes.kitchen_sink()
    finally:
        for k, v in old.items():
            setattr(scenes, k, v)
    compiled = compiler.compile_scene(full)
    lights = types.SimpleNamespace(root=full.root, light_nodes=[n for n in full.root.levelorder() if getattr(n, "light", None) is not None])
    n, max_events, record_every = 900, 48, 3
    np.random.seed(77)
    pos, direc, wl, sources = emit.emit_bundle(lights, n)
    data = O.reference_trace_bundle(compiled, pos, direc, wl, 31, 1000, max_events, 2, 1, record_every)
    result = api.EngineResult(compiled, data, sources, max_events, record_every, 0.0)
    out = {f"data/{k}": np.asarray(v) for k, v in data.items()}
    out["sources"] = np.array(sources)
    out["par"] = np.array([n, max_events, record_every], dtype=np.int64)
    for k, v in describe_engine_result(result, api.MOMENT_PROPERTIES).items():
        out[f"ref/{k}"] = v
    # ... and the reference's pure-Python tally of the same histories (engine/tally.py:86-150), recorder by recorder
    tally = ref_module("pvtrace.engine.tally").tally_histories(full, list(result.histories()))
    for k, v in describe_recorders(tally, api.MOMENT_PROPERTIES).items():
        out[f"tally/{k}"] = v
    save("engine_result.npz", **out)


def make_recorder_ids():
    """The reference's recorder vocabulary (engine/recorder.py:33-55: PROPERTIES, EVENTS) and what its constructors
    refuse, as JSON."""
    import json

    ref_module("pvtrace.data.lumogen_f_red_305")   # (makes sure the top-level namespace stub exists)
    if "pvtrace.engine" not in sys.modules:   # the engine package's __init__ imports its compiler, which needs anytree:
        pkg = types.ModuleType("pvtrace.engine")   # bypassed like the top-level __init__ (recorder.py imports nothing)
        pkg.__path__ = [os.path.join(REF, "engine")]
        sys.modules["pvtrace.engine"] = pkg
    rec = ref_module("pvtrace.engine.recorder")

    def refuses(fn):
        try:
            fn()
        except ValueError as exc:
            return str(exc)
        return None

    doc = {
        "PROPERTIES": dict(rec.PROPERTIES), "EVENTS": dict(rec.EVENTS),
        "refused": {
            "histogram_unknown_property": refuses(lambda: rec.Histogram("colour", 0, 1, 4)),
            "histogram_empty_range": refuses(lambda: rec.Histogram("x", 1, 1, 4)),
            "histogram_no_bins": refuses(lambda: rec.Histogram("x", 0, 1, 0)),
            "recorder_unknown_event": refuses(lambda: rec.Recorder("r", event="vanished")),
            "recorder_bad_histogram": refuses(lambda: rec.Recorder("r", histograms=[3])),
        },
        "defaults": {"event": rec.Recorder("r").event, "atol": rec.Recorder("r").atol, "facet": rec.Recorder("r").facet},
    }
    path = os.path.join(HERE, "recorder_ids.json")
    with open(path, "w") as fp:
        json.dump(doc, fp, indent=1, sort_keys=True)
    print("recorder_ids.json")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures can only be regenerated in the build container")
    if len(sys.argv) > 1 and sys.argv[1] == "--units":   # only the small unit fixtures added in round 5
        make_phase()
        make_surface()
        make_recorder_ids()
        make_py_tracer()
        make_emit()
        make_lsc_delegates()
        make_lsc_scenes()
        make_lsc_tracer()
        make_cfg5_tracer()
        make_object_methods()
        make_compiled_tables()
        make_engine_result()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--only":   # the named generators, e.g. --only make_lsc_tracer
        for fn in sys.argv[2:]:
            globals()[fn]()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--tallies":   # only the named config tallies (e.g. --tallies tiles6)
        make_config_tallies(only=sys.argv[2:])
        sys.exit(0)
    make_spectra()
    make_optics()
    make_geometry()
    make_transforms()
    make_phase()
    make_surface()
    make_recorder_ids()
    make_py_tracer()
    make_emit()
    make_lsc_delegates()
    make_lsc_scenes()
    make_lsc_tracer()
    make_cfg5_tracer()
    make_object_methods()
    make_compiled_tables()
    make_engine_result()
    make_traces()
    make_lsc_tallies()
    make_config_tallies()
    make_hist_spectra()


