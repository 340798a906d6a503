#!/usr/bin/env python3
"""Writes tests/golden/cli_text.json: the console text of the compiled reference (oracle/_ref/rt_ref, built from
/root/reference/src/main.cpp by `make -C oracle ref`) for a set of command lines that end in -x ("Exit without
rendering", src/main.cpp:479-483), i.e. its option parsing and its verbose block (src/main.cpp:352-477).
The CLI driver of the product (ray_tracing_amd/cli/raytracer.cpp) must print the same text for the same words.
Test infrastructure; run here, where the reference exists, never on the GPU box."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [
    ["-x"],
    ["-v", "-x"],
    ["-vVx"],
    ["-v", "-x", "-w", "64", "--height=32", "-r", "-s", "4"],
    ["-v", "-x", "-w", "1024", "-h", "1024", "-s", "1024", "-S", "1", "-r", "50", "-t", "16", "-o", "big.png"],
    ["-v", "-x", "-w", "1200", "-h", "800", "-s", "64", "-r", "8", "-p", "-m"],
    ["-v", "-x", "--width", "300", "--height", "200", "--samples", "7", "--sub-samples", "3", "--recurse", "5",
     "--threads", "2", "--tw", "8", "--th", "4", "--rx", "1", "--ry", "2", "--rw", "3", "--rh", "4", "--windowed",
     "--input", "in.txt", "--output=o.png", "--passes", "--multi", "--extra-verbose"],
    ["-v", "-x", "-r", "7", "stray", "-w", "10"],
    ["-v", "-x", "-w", "4000", "-h", "4000", "-s", "1"],
    ["-xv", "-h", "2", "-w", "2", "-s", "1"],
]


def main():
    exe = os.path.join(HERE, "_ref", "rt_ref")
    out = []
    for args in CASES:
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=60, cwd="/tmp")
        assert p.returncode == 0, (args, p.stderr)
        out.append({"args": args, "stdout": p.stdout})
    path = os.path.join(HERE, "..", "tests", "golden", "cli_text.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.normpath(path), len(out), "cases")


if __name__ == "__main__":
    main()
