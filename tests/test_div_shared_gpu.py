"""The culled scans and Russian roulette divide several numerators by one denominator (both roots of every sphere by
|d|^2, objects.go:55-59; the three attenuation components by the roulette probability, renderer.go:390) and share the
reciprocal refinement of the IEEE division between them.  That is only legal if every quotient keeps the bits of the
plain division, which the Go code performs: checked here on 4*10^9 operand pairs on the GPU."""
import pytest

pytestmark = pytest.mark.gpu


def test_shared_reciprocal_division_is_the_ieee_division(gpu_ctx):
    from path_trace_golang_amd import capi

    L = capi.load()
    total = 0
    for seed in (1, 2, 3, 4):
        bad = L.pt_debug_div_selftest(gpu_ctx.handle, 1000, seed)
        assert bad == 0, "%d of 10^9 quotients differ from the IEEE division (seed %d)" % (bad, seed)
        total += 1000
    assert total == 4000
