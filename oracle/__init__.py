"""CPU oracle of the ECoFLaP hot path — TEST INFRASTRUCTURE, not the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker / timed CPU baseline.  Parity status:
PINNED against golden vectors generated from the reference's own Python pruners
(tests/golden/make_golden.py; tests/test_oracle_golden.py).
"""
from .binding import Oracle, build, load  # noqa: F401
from .allocator import compute_sparsity_per_group, torch_sum_f32  # noqa: F401
