"""bench.py on a variant build of the library (A/B runs on one box): PF_LIB_VARIANT=<path> python tools/bench_variant.py [bench args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointfoot_amd import _abi
if os.environ.get("PF_LIB_VARIANT"):
    _abi.lib_path = lambda: os.path.abspath(os.environ["PF_LIB_VARIANT"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
