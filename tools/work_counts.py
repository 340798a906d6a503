#!/usr/bin/env python3
"""Per-path work counts of the Cornell box from the instrumented CPU oracle (what SURVEY.md 3.3 measured on
the reference), and the canonical FLOP count per path that bench.py's roofline uses (SURVEY.md 8d costs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from ray_tracing_amd.api import HostScene
hs = HostScene("cornell_box", 1.0)
O.counters(reset=True)
O.render(hs, 256, 256, 0, 16, 1, 50, O.ORC_RNG_COUNTER, 0, 8)
c = O.counters()
p = c["paths"]
per = {k: v / p for k, v in c.items()}
print({k: round(v, 3) for k, v in per.items()})
# SURVEY.md 8d canonical costs: slab 15; rect reject 6 (+12 on accept); sphere 23 (+15 on accept);
# diffuse shade 100; specular 40; ray-gen 20.  Accept counts: one accepted primitive per hit ray.
hits = per["rays"] - per["end_miss"]
flop = per["slab_tests"] * 15 + per["rect_tests"] * 6 + per["sphere_tests"] * 23 + hits * 12 + per["scatters"] * 85 + 20
print(f"canonical FLOP per path ~ {flop:.0f}")
