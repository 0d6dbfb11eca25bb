"""BASELINE configs[4] with DISTINCT reads on one MI355X -- launcher.  The run itself (generator -> index -> traversal -> phase 4, every
stage against a checker, the CPU oracle among them) is test tooling and lives in tests/config5_distinct_reads.py; this file only starts it:

    gpurun --timeout 1200 -- 'python scripts/distinct_151g.py > gpurun_out/distinct151.log 2>&1'
    (python scripts/distinct_151g.py 50000000 for a rehearsal at config 2's size)
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
runpy.run_module("tests.config5_distinct_reads", run_name="__main__")
