#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.  Run HERE (the build container), where
numpy 2.2.6 / scikit-learn 1.7.2 / scipy 1.15.3 are importable; the GPU box only reads the
committed files.  Nothing from /root/reference is copied: fixtures hold numbers only.

  dbscan_*.npz      inputs (float32) + labels / core mask of the REAL sklearn call the reference
                    makes (utils/tower_extraction.py:107-112)
  numpy_stats.json  np.mean(axis=0) (float32, sequential) and np.percentile(z,25) of seeded
                    EPSG:4547-scale arrays (utils/tower_extraction.py:63,83)
  kuangxuan_boxes.json  boxes of the reference's own example tower (ui/extract.py:460-464);
                    when /root/reference is present they are produced by importing the
                    reference's ui/extract.py (open3d / laspy replaced by empty placeholder
                    modules - its kuangxuan path is pure numpy), otherwise the values recorded
                    in SURVEY.md section 8c are written.
  e2e_config1.npz   self-golden (flagged as such): oracle end-to-end result for BASELINE
                    config 1 (1 M-pt corridor, 3 towers)
  refrun_*.npz      what the REFERENCE'S OWN utils/tower_extraction.py::extract_towers produced
                    when executed here on seeded inputs (see reference_runs() below): its log
                    lines, progress values, every cluster_points array it handed to trimesh
                    (as per-cluster SHA-256 + the label vector they imply), the tower dicts it
                    returned and the coordinates it handed to laspy for the per-tower files.
                    The modules the reference imports but this image lacks are replaced by
                    recorders: laspy (array-backed reader, recording writer), trimesh
                    (PointCloud records the points; its bounding_box_oriented is
                    oracle/obb.py - stage D1 stays PARITY UNPINNED), pandas (records the
                    DataFrame rows), open3d (empty).  numpy and scikit-learn are the real
                    libraries, called by the reference's own statements.
"""
import json
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def dbscan_cases():
    from sklearn.cluster import DBSCAN
    rng = np.random.default_rng(20250829)

    def blobs(n, k, spread, sigma, clutter):
        per = (n - clutter) // k
        parts = [rng.normal(rng.uniform(-spread, spread, 3), sigma, (per, 3)) for _ in range(k)]
        parts.append(rng.uniform(-spread * 1.3, spread * 1.3, (n - per * k, 3)))
        X = np.vstack(parts).astype(np.float32)
        return X[rng.permutation(len(X))]

    def towers(n, ntow, length):
        per = int(n * 0.85) // ntow
        parts = [rng.normal([(t + 0.5) * length / ntow, 50.0, 22.0], [2.5, 2.5, 9.0], (per, 3))
                 for t in range(ntow)]
        rest = n - per * ntow
        parts.append(np.column_stack([rng.uniform(0, length, rest), rng.uniform(0, 100, rest),
                                      rng.uniform(0, 30, rest)]))
        X = np.vstack(parts).astype(np.float32)
        return X[rng.permutation(len(X))]

    cases = {
        "blobs600": (blobs(600, 3, 6.0, 1.0, 150), 1.0, 8, 0),
        "towers5000": (towers(5000, 3, 900.0), 8.0, 80, 0),
        "towers30000_chunk10000": (towers(30000, 5, 1500.0), 8.0, 80, 10000),
        "all_noise": (rng.uniform(0, 1000, (2000, 3)).astype(np.float32), 8.0, 80, 0),
    }
    # border tie: one point within eps of two clusters (sklearn gives it to the first cluster)
    a = np.column_stack([np.linspace(0, 1, 30), np.zeros(30), np.zeros(30)])
    b = np.column_stack([np.linspace(3.2, 4.2, 30), np.zeros(30), np.zeros(30)])
    cases["border_tie"] = (np.vstack([b, [[2.1, 0, 0]], a]).astype(np.float32), 1.15, 8, 0)
    for name, (X, eps, ms, chunk) in cases.items():
        cs = chunk if chunk else len(X)
        labels = np.full(len(X), -1, np.int32)
        core = np.zeros(len(X), np.uint8)
        cur = 0
        for s in range(0, len(X), cs):
            cl = DBSCAN(eps=eps, min_samples=ms, n_jobs=-1, algorithm="ball_tree").fit(X[s:s + cs])
            lab = cl.labels_.copy()
            lab[lab != -1] += cur
            labels[s:s + cs] = lab
            core[s + cl.core_sample_indices_] = 1
            if (lab != -1).any():
                cur = lab.max() + 1
        np.savez_compressed(os.path.join(HERE, f"dbscan_{name}.npz"), X=X, labels=labels, core=core,
                            eps=eps, min_samples=ms, chunk=chunk)
        print(name, len(X), "clusters", labels.max() + 1, "noise", (labels == -1).sum())


def numpy_stats():
    out = {}
    for n in (1000, 1_000_000, 2_000_000):
        rng = np.random.default_rng(20250829 + n)
        raw = (rng.random((n, 3)) * [1000.0, 100.0, 30.0] + [437000.0, 3139000.0, 80.0]).astype(np.float32)
        c = np.mean(raw, axis=0)
        z = raw[:, 2] - c[2]
        p = np.percentile(z, 25)
        out[str(n)] = dict(seed=20250829 + n, checksum=int(raw.view(np.uint32).sum(dtype=np.uint64)),
                           centroid_bits=[int(v) for v in c.view(np.uint32)],
                           pct25_bits=int(np.float32(p).view(np.uint32)),
                           centroid=[float(v) for v in c], pct25=float(p))
    print("numpy_stats", {k: v["centroid"] for k, v in out.items()})
    out["_numpy"] = {"version": np.__version__, "major": int(np.__version__.split(".")[0]),
                     "note": "np.percentile's index arithmetic is float32 under numpy >= 2 (NEP 50); the device select mirrors that"}
    json.dump(out, open(os.path.join(HERE, "numpy_stats.json"), "w"), indent=1)


def kuangxuan_boxes():
    center = [437587.898, 3140691.58, 131.457]
    extent = [20.1, 20.1, 17.4]
    out = {"center": center, "extent": extent, "source": None, "presets": {}}
    ref = "/root/reference"
    try:
        if not os.path.isdir(ref):
            raise RuntimeError("no reference tree")
        for name in ("open3d", "laspy"):
            sys.modules.setdefault(name, types.ModuleType(name))
        sys.path.insert(0, ref)
        import importlib
        ext = importlib.import_module("ui.extract")
        for preset in ("kuangxuan_original", "kuangxuan_conservative", "kuangxuan_aggressive"):
            method, params = ext.get_bbox_preset(preset)
            lo, hi = ext.create_bbox_using_kuangxuan_method(np.array(center), 20.1, 17.4, **params)
            pts, _ = ext.create_bbox_lineset_from_bounds(lo, hi)
            out["presets"][preset] = dict(min=[float(v) for v in lo], max=[float(v) for v in hi],
                                          lines=np.asarray(pts).tolist())
        out["source"] = "reference ui/extract.py executed in the build container"
        sys.path.remove(ref)
        for name in [m for m in sys.modules if m == "ui" or m.startswith("ui.")]:
            del sys.modules[name]
    except Exception as e:                                  # SURVEY.md section 8c values
        out["source"] = f"SURVEY.md section 8c (reference import not possible: {e})"
        out["presets"] = {
            "kuangxuan_original": dict(min=[437567.798, 3140681.53, 114.057], max=[437621.465, 3140711.68, 166.257]),
            "kuangxuan_conservative": dict(min=[437571.818, 3140683.54, 122.757], max=[437612.018, 3140707.66, 157.557]),
            "kuangxuan_aggressive": dict(min=[437557.748, 3140675.50, 105.357], max=[437628.098, 3140721.73, 183.657]),
        }
    json.dump(out, open(os.path.join(HERE, "kuangxuan_boxes.json"), "w"), indent=1)
    print("kuangxuan:", out["source"])


def e2e_config1():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    from oracle import towers as ot
    from pointcloudhookup_amd import synth
    pts = synth.corridor_numpy(1_000_000, seed=synth.SEED0 + 0, kind="corridor", offset=True, towers=3)
    res = {}
    for order in ("unsorted", "trimesh_sorted"):
        r = ot.extract_towers_arrays(pts[:, 0], pts[:, 1], pts[:, 2], fit="c", extent_order=order)
        res[order] = r
        print(order, "filtered", len(r["ground"]["filtered"]), "candidates", r["n_candidates"],
              "towers", [(t["label"], round(t["height"], 2), round(t["width"], 2)) for t in r["towers"]])
    r = res["unsorted"]
    np.savez_compressed(
        os.path.join(HERE, "e2e_config1.npz"),
        note="self-golden: produced by the repo's own CPU oracle, not by the reference",
        seed=synth.SEED0, n=1_000_000,
        centroid=r["ground"]["centroid"], base=r["ground"]["base"], threshold=r["ground"]["threshold"],
        n_filtered=len(r["ground"]["filtered"]), labels=r["labels"].astype(np.int16),
        n_candidates=r["n_candidates"],
        **{f"{o}_{k}": np.array([t[k] for t in res[o]["towers"]], dtype=np.float64).reshape(len(res[o]["towers"]), -1)
           for o in res for k in ("center", "extent", "north_angle", "label")})


# ---------------------------------------------------------------------------------------------
# Reference-run fixtures: the reference's own extract_towers executed under recording modules.
REF = "/root/reference"
REFRUN_SCALES = np.array([0.001, 0.001, 0.001])
REFRUN_OFFSETS = np.array([437000.0, 3139000.0, 0.0])


def refrun_inputs(name):
    """Seeded inputs of the reference-run cases.  Returns (x, y, z float64 as laspy would hand
    them over, XYZ int32 | None, extract_towers kwargs).  Shared with the tests, which rebuild
    the same inputs and check the checksum stored in the fixture."""
    from pointcloudhookup_amd import synth

    def quantise(pts):
        XYZ = np.round((pts - REFRUN_OFFSETS) / REFRUN_SCALES).astype(np.int32)
        cols = [XYZ[:, a].astype(np.float64) * REFRUN_SCALES[a] + REFRUN_OFFSETS[a] for a in range(3)]
        return cols[0], cols[1], cols[2], XYZ

    if name == "config1_1m":                     # BASELINE config 1: 1 M pts, 3 towers
        pts = synth.corridor_numpy(1_000_000, seed=synth.SEED0, kind="corridor", offset=True, towers=3)
        return (*quantise(pts), {})
    if name == "towers5x3":                     # 5 towers, several 50k chunks cut through towers
        rng = np.random.default_rng(77)
        g = np.column_stack([rng.uniform(0, 300, 300_000), rng.uniform(0, 100, 300_000), rng.normal(0, 0.05, 300_000)])
        tw = [np.column_stack([rng.normal(30 + 60 * t, 2.5, 24_000), rng.normal(50, 2.5, 24_000),
                               np.clip(rng.normal(22, 9, 24_000), 0.5, 45)]) for t in range(5)]
        clutter = np.column_stack([rng.uniform(0, 300, 3000), rng.uniform(0, 100, 3000), rng.uniform(3, 40, 3000)])
        pts = np.vstack([g] + tw + [clutter]) + synth.GLOBAL_OFFSET
        pts = pts[rng.permutation(len(pts))]
        return (*quantise(pts), {})
    if name == "fallback":                       # < 1000 survivors at +3.0 -> threshold +1.0 (:87-89)
        rng = np.random.default_rng(501)
        g = np.column_stack([rng.uniform(0, 300, 20000), rng.uniform(0, 100, 20000), rng.normal(0, 0.05, 20000)])
        mid = np.column_stack([rng.normal(150, 2.0, 2500), rng.normal(50, 2.0, 2500), rng.uniform(1.2, 2.9, 2500)])
        top = np.column_stack([rng.normal(150, 2.0, 600), rng.normal(50, 2.0, 600), rng.uniform(3.5, 25, 600)])
        far = np.column_stack([rng.uniform(0, 300, 400), rng.uniform(0, 100, 400), rng.uniform(1.5, 2.5, 400)])
        pts = np.vstack([g, mid, top, far]) + synth.GLOBAL_OFFSET
        pts = pts[rng.permutation(len(pts))]
        return (*quantise(pts), dict(eps=3.0, min_points=20, min_height=5.0, min_width=2, aspect_ratio_threshold=0.3))
    if name == "nonfinite":                      # one NaN x: the centroid's x is NaN, every chunk fails in sklearn
        pts = synth.corridor_numpy(300_000, seed=synth.SEED0 + 3, kind="corridor", offset=True, towers=3)
        x, y, z, _ = quantise(pts)
        x = x.copy()
        x[123456] = np.nan
        return x, y, z, None, {}
    if name == "empty":                          # zero points: np.percentile raises, the filter stage returns []
        e = np.zeros(0)
        return e, e, e, np.zeros((0, 3), np.int32), {}
    raise KeyError(name)


REFRUN_CASES = ["config1_1m", "towers5x3", "fallback", "nonfinite", "empty"]


def input_checksum(x, y, z):
    import hashlib
    h = hashlib.sha256()
    for a in (x, y, z):
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _recording_modules(registry, rec):
    """Stand-ins for the modules the reference imports that this image lacks.  They hold no
    algorithm of the hot path except trimesh's box fit, which is oracle/obb.py (unpinned)."""
    from oracle import obb as oobb

    laspy = types.ModuleType("laspy")

    class _Header:
        def __init__(self, point_format=3, version=(1, 2)):
            self.point_format, self.version = point_format, version
            self.scales, self.offsets = REFRUN_SCALES.copy(), REFRUN_OFFSETS.copy()

    class _Las:
        def __init__(self, header):
            self.header = header

        def write(self, path):
            rec["las_writes"].append(dict(path=str(path), x=np.array(self.x), y=np.array(self.y),
                                          z=np.array(self.z), scales=np.array(self.header.scales),
                                          offsets=np.array(self.header.offsets)))

    class _Reader:
        def __init__(self, path):
            self.path = path

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def read(self):
            x, y, z = registry[self.path]
            las = _Las(_Header())
            las.x, las.y, las.z = x, y, z
            return las

    laspy.open = lambda path, *a, **k: _Reader(path)
    laspy.LasHeader = _Header
    laspy.LasData = _Las

    trimesh = types.ModuleType("trimesh")

    class _Box:
        def __init__(self, extents, transform):
            self.extents, self.transform = extents, transform

    class PointCloud:
        def __init__(self, points):
            self.points = points
            rec["cluster_points"].append(np.array(points))

        @property
        def bounding_box_oriented(self):
            ext, tf = oobb.bounding_box_oriented(self.points, rec["extent_order"])
            return _Box(ext, tf)

    trimesh.PointCloud = PointCloud

    pandas = types.ModuleType("pandas")

    class DataFrame:
        def __init__(self, rows):
            self.rows = rows

        def to_excel(self, path, index=False):
            rec["xlsx_rows"] = self.rows
            rec["xlsx_path"] = str(path)

    pandas.DataFrame = DataFrame
    return {"laspy": laspy, "trimesh": trimesh, "pandas": pandas, "open3d": types.ModuleType("open3d")}


def reference_runs(which=None):
    """Executes /root/reference/utils/tower_extraction.py::extract_towers on the seeded cases and
    writes tests/golden/refrun_<case>.npz.  Build container only."""
    import importlib
    import shutil
    import tempfile
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present - reference-run fixtures can only be made in the build container")
    for case in (which or REFRUN_CASES):
        for order in ("unsorted", "trimesh_sorted"):
            x, y, z, XYZ, kwargs = refrun_inputs(case)
            registry = {"mem://" + case: (x, y, z)}
            rec = dict(cluster_points=[], las_writes=[], xlsx_rows=[], extent_order=order)
            saved = {k: sys.modules.get(k) for k in ("laspy", "trimesh", "pandas", "open3d")}
            sys.modules.update(_recording_modules(registry, rec))
            work = tempfile.mkdtemp(prefix="refrun_", dir=os.path.join(ROOT, "gpurun_out"))
            cwd = os.getcwd()
            sys.path.insert(0, REF)
            logs, prog = [], []
            try:
                os.chdir(work)
                te = importlib.import_module("utils.tower_extraction")
                raised = ""
                try:
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        towers = te.extract_towers("mem://" + case, progress_callback=prog.append,
                                                   log_callback=logs.append, **kwargs)
                except Exception as e:                      # the reference's own escape, recorded as such
                    raised, towers = f"{type(e).__name__}: {e}", []
            finally:
                os.chdir(cwd)
                sys.path.remove(REF)
                for k in [m for m in sys.modules if m == "utils" or m.startswith("utils.")]:
                    del sys.modules[k]
                for k, v in saved.items():
                    if v is None:
                        sys.modules.pop(k, None)
                    else:
                        sys.modules[k] = v
                shutil.rmtree(work, ignore_errors=True)
            if order == "unsorted":
                base = dict(rec=rec, logs=logs, prog=prog, towers=towers, raised=raised)
                cps = rec["cluster_points"]
                # the label vector the reference's masks imply: rebuild the filtered array with the
                # reference's own numpy statements and locate every handed-over cluster in it
                raw = np.stack([x, y, z], axis=1).astype(np.float32)
                with np.errstate(all="ignore"):
                    cen = np.mean(raw, axis=0)
                    pts = raw - cen
                n_f = None
                for line in logs:
                    if line.startswith("✅ 高度过滤完成"):
                        n_f = int(line.split(":")[-1])
                labels = None
                if cps:
                    zv = pts[:, 2]
                    bh = np.percentile(zv, 25)
                    filt = pts[zv > bh + 3.0]
                    if len(filt) < 1000:
                        filt = pts[zv > bh + 1.0]
                    labels = np.full(len(filt), -1, np.int32)
                    # float32 quantisation makes duplicate rows common; inside one 50 000-row chunk
                    # identical rows always share a label, so a cluster is "every row of its chunk
                    # whose value occurs in the handed-over array"; chunks are tried in order
                    row_t = [("", np.float32)] * 3
                    fv = np.ascontiguousarray(filt).view(row_t).ravel()
                    cs, c = 50000, 0
                    for k, cp in enumerate(cps):
                        cv = np.ascontiguousarray(cp.astype(np.float32)).view(row_t).ravel()
                        while True:
                            assert c * cs < len(filt), f"cluster {k} not found in any chunk"
                            seg = slice(c * cs, (c + 1) * cs)
                            rows = np.flatnonzero(np.isin(fv[seg], cv) & (labels[seg] == -1)) + c * cs
                            if len(rows) == len(cp) and np.array_equal(filt[rows], cp):
                                break
                            c += 1
                        labels[rows] = k
                    for k, cp in enumerate(cps):
                        assert np.array_equal(filt[labels == k], cp)
            else:
                other = dict(towers=towers, logs=logs)
        u, s = base, other
        tower_fields = {}
        for tag, tw in (("unsorted", u["towers"]), ("trimesh_sorted", s["towers"])):
            tower_fields[f"{tag}_center"] = np.array([t["center"] for t in tw], np.float64).reshape(len(tw), 3)
            tower_fields[f"{tag}_extent"] = np.array([t["extent"] for t in tw], np.float64).reshape(len(tw), 3)
            tower_fields[f"{tag}_rotation"] = np.array([t["rotation"] for t in tw], np.float64).reshape(len(tw), 3, 3)
            tower_fields[f"{tag}_north_angle"] = np.array([t["north_angle"] for t in tw], np.float64)
            tower_fields[f"{tag}_height"] = np.array([t["height"] for t in tw], np.float64)
            tower_fields[f"{tag}_width"] = np.array([t["width"] for t in tw], np.float64)
            tower_fields[f"{tag}_npoints"] = np.array([len(t["points"]) for t in tw], np.int64)
            tower_fields[f"{tag}_points_sha"] = np.array([_sha(t["points"]) for t in tw])
        tower_fields["trimesh_sorted_logs"] = np.array(s["logs"])
        writes = u["rec"]["las_writes"]
        sc, of = REFRUN_SCALES, REFRUN_OFFSETS
        np.savez_compressed(
            os.path.join(HERE, f"refrun_{case}.npz"),
            note="produced by executing the reference's utils/tower_extraction.py::extract_towers in the build "
                 "container under recording modules (see gen_golden.py); numbers only",
            case=case, n=len(x), input_sha=input_checksum(x, y, z), raised=u["raised"],
            kwargs_json=json.dumps(kwargs), scales=sc, offsets=of,
            logs=np.array(u["logs"]), progress=np.array(u["prog"], np.int64),
            n_filtered_logged=-1 if n_f is None else n_f,
            n_clusters=len(u["rec"]["cluster_points"]),
            cluster_sizes=np.array([len(c) for c in u["rec"]["cluster_points"]], np.int64),
            cluster_sha=np.array([_sha(c) for c in u["rec"]["cluster_points"]]),
            labels=np.zeros(0, np.int32) if labels is None else labels.astype(np.int16 if len(u["rec"]["cluster_points"]) < 32000 else np.int32),
            las_paths=np.array([os.path.basename(w["path"]) for w in writes]),
            las_XYZ_sha=np.array([_sha(np.stack([np.round((w[a] - of[i]) / sc[i]).astype(np.int32)
                                                 for i, a in enumerate("xyz")], axis=1)) for w in writes]),
            las_xyz_f64_sha=np.array([_sha(np.stack([w["x"], w["y"], w["z"]], axis=1)) for w in writes]),
            xlsx_ids=np.array([r["ID"] for r in u["rec"]["xlsx_rows"]]),
            xlsx_values=np.array([[r["经度"], r["纬度"], r["海拔高度"], r["杆塔高度"], r["北方向偏角"], r["宽度"], r["长宽比"]]
                                  for r in u["rec"]["xlsx_rows"]], np.float64).reshape(len(u["rec"]["xlsx_rows"]), 7),
            **tower_fields)
        print(case, "n", len(x), "filtered", n_f, "clusters", len(u["rec"]["cluster_points"]),
              "towers", len(u["towers"]), "/", len(s["towers"]), "logs", len(u["logs"]))


def gim_match_contract():
    """Runs the reference's OWN consumer of the tower dicts (utils/table_match_gim.py::match_towers) on the
    towers of refrun_config1_1m / refrun_towers5x3 and writes tests/golden/gim_match.json.  PyQt5 and pyproj
    are not installed: empty placeholder modules stand in (the module only needs their names at import time;
    its geoid transformer then runs in its documented fallback mode, ellipsoid height - 25 m)."""
    import contextlib
    import importlib
    import io
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present")

    class _Any:
        def __init__(self, *a, **k): pass
        def __getattr__(self, n): return _Any()
        def __call__(self, *a, **k): return _Any()

    def fake(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__getattr__ = lambda n: _Any
        return m

    def no_grid(*a, **k):
        raise RuntimeError("no geoid grid offline")

    stubs = {"PyQt5": fake("PyQt5"), "PyQt5.QtWidgets": fake("PyQt5.QtWidgets"), "PyQt5.QtCore": fake("PyQt5.QtCore"),
             "PyQt5.QtGui": fake("PyQt5.QtGui"),
             "pyproj": fake("pyproj", Transformer=types.SimpleNamespace(from_pipeline=no_grid),
                            datadir=types.SimpleNamespace(get_data_dir=lambda: "/nonexistent"))}
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    sys.path.insert(0, REF)
    out = {"note": "output of the reference's utils/table_match_gim.py::match_towers (fallback geoid mode) on the "
                   "towers of the refrun fixtures; transformer: lon = 112 + (x - 437000) * 1e-5, lat = 28 + (y - 3139000) * 9e-6",
           "cases": {}}
    try:
        tm = importlib.import_module("utils.table_match_gim")

        class T:
            def transform(self, x, y):
                return 112.0 + (x - 437000.0) * 1e-5, 28.0 + (y - 3139000.0) * 9e-6

        for case in ("config1_1m", "towers5x3"):
            g = np.load(os.path.join(HERE, f"refrun_{case}.npz"))
            towers = [dict(center=g["trimesh_sorted_center"][i], height=float(g["trimesh_sorted_height"][i]),
                           north_angle=float(g["trimesh_sorted_north_angle"][i]))
                      for i in range(len(g["trimesh_sorted_center"]))]
            tr = T()
            gim = []
            for k, t in enumerate(towers):                  # GIM towers: near the 1st, 60 m off the 2nd, 150 m too high for the 3rd ...
                lon, lat = tr.transform(t["center"][0], t["center"][1])
                dx = [5.0, 60.0, 10.0, 0.0, 49.0][k % 5]
                dh = [0.0, 0.0, 150.0, -99.0, 3.0][k % 5]
                gim.append({"lat": lat, "lng": lon + dx / 97000.0, "h": t["center"][2] - 25.0 + dh})
            with contextlib.redirect_stdout(io.StringIO()):
                matched, conv = tm.match_towers(gim, towers, tr)
            out["cases"][case] = {"gim": gim, "matched": [list(m) for m in matched],
                                  "converted": [{k: (list(map(float, v)) if hasattr(v, "__len__") and not isinstance(v, str) else v)
                                                 for k, v in c.items()} for c in conv]}
            print(case, "towers", len(towers), "matched", matched)
    finally:
        sys.path.remove(REF)
        for k in [m for m in sys.modules if m == "utils" or m.startswith("utils.")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    json.dump(out, open(os.path.join(HERE, "gim_match.json"), "w"), indent=1, default=float)


if __name__ == "__main__":
    which = sys.argv[1:] or ["dbscan", "numpy", "boxes", "e2e", "refrun", "gim"]
    if "dbscan" in which:
        dbscan_cases()
    if "numpy" in which:
        numpy_stats()
    if "boxes" in which:
        kuangxuan_boxes()
    if "e2e" in which:
        e2e_config1()
    if "gim" in which:
        gim_match_contract()
    if any(w == "refrun" or w.startswith("refrun:") for w in which):
        reference_runs([w.split(":", 1)[1] for w in which if w.startswith("refrun:")] or None)
