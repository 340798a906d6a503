"""Host-side tree builders (no GPU): the library's SAH tree with half-float boxes (rt_bvh.cpp) and the side table of the
ordered walk of the reference's tree (rt_order.cpp), checked by a small C++ program built here with g++."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ray_tracing_amd")
SCENES = ["cornell_box", "cornell_box_volumes", "random_spheres", "next_week_final", "simple_scene", "another_simple",
          "two_perlin_spheres", "sphere_field_4k"]


def test_trees_hold_every_primitive_inside_conservative_boxes(tmp_path):
    """Half-float boxes are rounded outwards (a box may be too large, never too small); every primitive is a leaf exactly
    once, list members each on their own; the depth stays within the per-lane stack; the reference's entry array is a
    binary tree (what world_hit_ordered needs)."""
    exe = str(tmp_path / "bvh_check")
    if not os.path.exists(os.path.join(PKG, "build", "rt_order.o")) or not os.path.exists(os.path.join(PKG, "librt_amd.so")):
        subprocess.run(["make", "-C", PKG], check=True)  # normally done by __graft_entry__.build()
    subprocess.run(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "bvh_check.cpp"), os.path.join(PKG, "build", "rt_order.o"),
                    "-L" + PKG, "-lrt_amd", "-Wl,-rpath," + PKG, "-o", exe], check=True)
    p = subprocess.run([exe] + SCENES, capture_output=True, text=True, timeout=600)
    print(p.stdout)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "half rounding: 0 bad" in p.stdout and "failures: 0" in p.stdout
    assert p.stdout.count("ok=1") == len(SCENES)
