"""TEST INFRASTRUCTURE ONLY -- the CPU checker for the cipher path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (modulate_amd) never does.
"""
from .oracle import *  # noqa: F401,F403
