"""Command-line front end of the randomised parity sweep (tests/fuzz_sweep.py; the -m gpu suite runs two seeds of it).
Usage: python tools/dev/fuzz_parity.py [n_cases] [seed] [dir_scale]      (dir_scale 0 = frames without flow: the hand-scheduled forward walk)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import fuzz_sweep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dir_scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
fails = fuzz_sweep.run(n, seed, dir_scale)
print("failures:", len(fails))
sys.exit(1 if fails else 0)
