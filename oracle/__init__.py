"""
ORACLE package -- test infrastructure only.

CPU restatement of the reference's hot-path algorithms (C in *.c, the filter
designer in filter_design.py).  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this package; the product
(lsp-dsp-units_amd/) never does.

The reference itself cannot be built in this environment: its arithmetic core
(lsp-dsp-lib 1.0.36) and lsp-common-lib/lsp-runtime-lib/lsp-lltl-lib/lsp-test-fw
are un-vendored (modules.mk:23-51, fetched by `make fetch`) and absent, so there
is no oracle/_ref build.  Each restated function cites the reference file:line
it follows and is pinned against the reference's own unit-test expectations
(tests/test_oracle_*.py).
"""
from .binding import *  # noqa: F401,F403
from . import spectral  # noqa: F401,E402
from . import delay  # noqa: F401,E402
from . import equalizer  # noqa: F401,E402
