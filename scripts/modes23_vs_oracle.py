"""Launcher of tests/modes23_vs_oracle.py (test tooling: it runs the CPU oracle as its checker, so it lives under tests/)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runpy.run_module("tests.modes23_vs_oracle", run_name="__main__")
