"""Device-code audit that needs no GPU: every s_barrier of libdslsph's gfx950 ISA has the wave's LDS queue drained
(`s_waitcnt lgkmcnt(0)`) in front of it.  See tools/isa_audit.py for the failure this guards against."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_every_barrier_has_the_lds_queue_drained():
    import isa_audit
    bad, total = isa_audit.unprotected_barriers(isa_audit.device_asm())
    assert total > 100  # (the tiled kernels alone hold that many)
    assert not bad, bad[:3]
