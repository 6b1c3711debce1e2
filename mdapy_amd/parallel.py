"""Thread-count plumbing kept for signature compatibility with the reference
(src/mdapy/parallel.py:1-53, MDAPY_NUM_THREADS in src/mdapy/__init__.py:14-33).
Every native function of the reference takes a trailing ``num_t``; the HIP
kernels ignore it."""
import os


def get_num_threads() -> int:
    v = os.environ.get("MDAPY_NUM_THREADS")
    if v is not None:
        try:
            n = int(v)
            if n > 0:
                return n
        except ValueError:
            pass
    return os.cpu_count() or 1
