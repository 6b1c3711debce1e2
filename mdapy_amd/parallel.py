"""Thread count of the reference's native functions (src/mdapy/parallel.py:1-53, MDAPY_NUM_THREADS in
src/mdapy/__init__.py:14-33).  Every one of them takes a trailing ``num_t``; the shims accept it for signature
compatibility and the HIP kernels have no use for it."""
import os


def get_num_threads():
    asked = os.environ.get("MDAPY_NUM_THREADS", "")
    if asked.strip().lstrip("+").isdigit() and int(asked) > 0:
        return int(asked)
    return os.cpu_count() or 1
