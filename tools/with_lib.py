"""Run a python script against ANOTHER build of the C-ABI library (same-box A/B of a kernel change):
    python tools/with_lib.py path/to/libother.so script.py [args ...]
Development tool: the product always loads centerpose_amd/libcenterpose_hip.so."""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from centerpose_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(sys.argv[0])))
runpy.run_path(sys.argv[0], run_name="__main__")
