#!/usr/bin/env python3
"""Time the triangle scenes (BASELINE config 4 and test.scn) — development aid; SKR_LIBRARY picks the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.time_configs import run

if __name__ == "__main__":
    print(os.environ.get("SKR_LIBRARY", "lib/libskr.so"), "SKR_NO_CULL" in os.environ and "no cull" or "")
    run("dragon.scn", 1920, 1080, reps=4, gillum=16)
    run("dragon.scn", 640, 480, reps=4)
    run("test.scn", 640, 360, reps=4, gillum=4, shadow=True)
