"""Launcher of tests/fullsize_check.py (test tooling: it runs the CPU oracle as its checker, so it lives under tests/)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runpy.run_module("tests.fullsize_check", run_name="__main__")
