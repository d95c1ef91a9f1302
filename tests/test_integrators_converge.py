"""The reference's own validation method: every integrator renders the same image (CMakeLists.txt:95-143 `make pt`, `make lt`,
...; SURVEY.md section 4 -- the reference has no other test).  Path tracing (algorithm_pt.cc:125-160: radiance transport,
SampleLight, emitted radiance collected at light hits) and light tracing (algorithm_lt.cc:125-163: importance transport,
SampleImportance, sensor response at aperture hits) are two different estimators of one image; on the device they share the
primitives, materials and samplers but differ in ray generation (LightSet::GenerateRay vs the thin lens), in the refraction
adjoint (ior^2 scaling) and in how a path reaches the sensor (Lens::Response).  Rows no hash can pin -- Disk as geometry
and as a light, Cylinder and triangle lights, lt itself -- are checked here the way the reference checks everything: the
two images must agree within their Monte-Carlo error.

The scene uses only configurations in which the REFERENCE'S OWN arithmetic lets the two converge.  Three reference
properties (restated faithfully by oracle and engine, EXPERIMENTS.md "the reference's own validation method") break
the agreement and are pinned by the second test instead of being "fixed":
  * prelude::SphereSA (sampling.h:185-199) returns (cos t cos p, cos t sin p, sin t): only the z >= 0 half of the sphere,
    non-uniformly -- a sphere LIGHT emits differently in lt than pt sees it;
  * kEPS = 1e-6 (constants.h:26) is below the rounding of a hit point on a curved surface: rays that leave a CYLINDER
    nearly parallel to its axis re-hit it at t ~ 1e-6 .. 1e-5 (primitive_cylinder.cc:100-142 with a = |d_perp|^2 small),
    so pt -- whose rays leave a lit cylinder TOWARDS the light above -- loses light that lt does not.
"""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENE = dict(
    materials=[(4, (30.0, 20.0, 10.0), 0.0), (0, (0.7, 0.6, 0.5), 0.0), (2, (0.8, 0.8, 0.8), 0.0), (3, (1.0, 1.0, 1.0), 1.5), (4, (5.0, 5.0, 9.0), 0.0),
               (1, (0.6, 0.6, 0.6), 20.0)],
    objects=[
        (2, 0, [0.0, 1.5, 0.0, 0.0, -1.0, 0.0, 0.6]),                      # disk light
        (0, 4, [-1.0, 1.4, -1.0, -1.0, 1.4, 1.0, -0.5, 1.4, 0.0]),        # triangle light
        (3, 4, [-0.95, -0.5, 0.2, 0.0, 1.0, 0.0, 0.08, 0.6]),             # cylinder light (in view, left)
        (2, 4, [0.9, 0.1, -0.6, -0.6, 0.0, 0.8, 0.12]),                    # small disk light facing the camera side
        (0, 1, [-3, -1, -3, 3, -1, 3, 3, -1, -3]), (0, 1, [-3, -1, -3, -3, -1, 3, 3, -1, 3]),   # floor
        (1, 2, [0.7, -0.6, -0.3, 0.4]), (1, 3, [0.0, -0.5, 0.8, 0.45]),   # mirror and glass spheres
        (1, 1, [-0.55, -0.65, -0.9, 0.35]),                                # diffuse sphere
        (2, 1, [0.25, -0.3, -1.6, -0.3, 0.3, 0.9, 0.6]),                  # diffuse disk
    ],
    transform=[1, 0, 0, 0, 0, 1, 0, 0.2, 0, 0, 1, 2.6, 0, 0, 0, 1], focal_length=0.05, focus_distance=2.6, radius=0.45, n_blades=5,
)
W, H = 24, 18


def block_stats(amber, scene, K=6, pt_spp=131072, lt_passes=400000):
    """K independent renders per integrator (seeds differ); 3 x 4-pixel blocks of the lower 12 rows: means, their ratio and
    the z score of the difference against the batch-to-batch standard error."""
    hs = amber.HostScene.create(**scene)
    sensor = amber.Sensor.default(W, H)
    pt = np.stack([hs.render(sensor, pt_spp, seed=1000 + k, samples_per_launch=65536, algorithm="pt")[0] for k in range(K)])
    lt = np.stack([hs.render(sensor, lt_passes, seed=2000 + k, samples_per_launch=20000, algorithm="lt")[0] for k in range(K)])
    assert np.isfinite(pt).all() and np.isfinite(lt).all()

    def blocks(a):
        return a[:, 6:18].reshape(a.shape[0], 4, 3, 6, 4, 3).sum(axis=(2, 4, 5))
    bp, bl = blocks(pt), blocks(lt)
    se = np.sqrt(bp.var(0, ddof=1) / K + bl.var(0, ddof=1) / K)
    mp, ml = bp.mean(0), bl.mean(0)
    coarse = lambda a: a.mean(0)[6:18].reshape(2, 6, 3, 8, 3).sum(axis=(1, 3, 4))          # 2 x 3 regions of 6 x 8 pixels
    return dict(pt=pt.mean(0), lt=lt.mean(0), bp=mp, bl=ml, z=(mp - ml) / np.maximum(se, 1e-30), ratio=ml / np.maximum(mp, 1e-30),
                sum_ratio=float(lt.mean(0).sum() / pt.mean(0).sum()), coarse_ratio=coarse(lt) / np.maximum(coarse(pt), 1e-30))


def test_path_tracing_and_light_tracing_converge_to_the_same_image(amber):
    """Agreement of the two estimators on total energy, on six image regions and pixel by pixel (correlation).  lt's
    per-pixel distribution is heavy-tailed (rare large splats where a surface is seen at a grazing angle), so its sample
    mean approaches the common limit from below; thresholds carry that margin (measured: total 1.00-1.03, regions within
    8 %, correlation 0.99 at 7.9e5 pt samples and 1.0e9 light paths per image)."""
    r = block_stats(amber, SCENE)
    print("lt/pt total", r["sum_ratio"], "regions", r["coarse_ratio"].round(3).tolist())
    assert 0.95 < r["sum_ratio"] < 1.05, r["sum_ratio"]
    assert np.corrcoef(r["pt"].ravel(), r["lt"].ravel())[0, 1] > 0.985
    assert np.abs(r["coarse_ratio"] - 1).max() < 0.12, r["coarse_ratio"]
    assert ((r["pt"] > 0) == (r["lt"] > 0)).mean() > 0.97                          # the same pixels are lit


def test_reference_properties_that_break_the_agreement_are_reproduced_not_fixed(amber):
    floor = SCENE["objects"][4:6]
    sc = copy.deepcopy(SCENE)
    sc["objects"] = [(1, 0, [1.2, 0.8, 0.0, 0.15])] + floor                        # a sphere LIGHT: SphereSA's half sphere
    r = block_stats(amber, sc, K=3, pt_spp=32768, lt_passes=100000)
    assert r["sum_ratio"] < 0.2, r["sum_ratio"]                                    # measured 0.054
    sc["objects"] = [SCENE["objects"][0], (3, 1, [-0.6, -0.8, -0.8, 0.0, 1.0, 0.0, 0.3, 0.9])]   # disk light over a diffuse CYLINDER
    r = block_stats(amber, sc, K=3, pt_spp=32768, lt_passes=100000)
    assert r["sum_ratio"] > 1.25, r["sum_ratio"]                                   # measured 1.43 (a 48-sided prism of triangles: 1.02)
