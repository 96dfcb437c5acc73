import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library gives a graph whose whole search is less work than one launch a single root chunk.  The tests
# run on small graphs and exist to exercise chunk tables, shared cursors and gather rounds, so they lift that
# floor; the full-size configuration tests put the default back.
os.environ.setdefault("GPM_MIN_CHUNK_COST", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
