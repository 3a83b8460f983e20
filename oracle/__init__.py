"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see kmsr_oracle.c).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.  The product
package (kmsr_amd) never imports this.
"""
from .oracle import *  # noqa: F401,F403
