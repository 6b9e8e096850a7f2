"""The decoder-step dispatch matrix as data: which kernel family each (live rows, caller) pair runs on
(sc_decoder_step_family evaluates the predicates the stages themselves use - model_decoder.hip: decoder_step_family).  Three
families are live: 1 general (split-K skinny products / tiled GEMMs: the fallback for geometries the packed kernels do not
take and for more than 512 live rows), 2 packed-fragment products (k_dstep.hip: the streaming decoder's step with its
p_choose hook, the v1 unit decoder's beam search - no packed embedding), 3 / 4 row-group products (k_dstep3.hip: greedy text
generation, beam search; 4 = cut into row groups above 64 rows).  A silent change of a rung would show here."""
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu

GREEDY, BEAM, STREAM, UNIT_BEAM, FORCED = 0, 1, 2, 3, 4


def _family(hip, rows, caller):
    return int(hip.lib.sc_decoder_step_family(hip.handle, rows, caller))


def test_dispatch_matrix_tiny_v2():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    hip = common.make_hip_streaming()
    assert [_family(hip, r, GREEDY) for r in (1, 2, 31, 32, 33, 64)] == [3] * 6      # row-group chain up to 64 rows
    assert _family(hip, 65, GREEDY) == 1 and _family(hip, 512, GREEDY) == 1          # greedy above 64 rows: general path
    assert [_family(hip, r, FORCED) for r in (1, 64)] == [3, 3] and _family(hip, 65, FORCED) == 1
    assert [_family(hip, r, BEAM) for r in (5, 60, 64)] == [3, 3, 3]                   # beams x utterances = live rows
    assert [_family(hip, r, BEAM) for r in (65, 80, 320, 512)] == [4] * 4             # wide row-group chain
    assert _family(hip, 513, BEAM) == 1                                                # beyond the wide chain: tiled GEMMs
    assert _family(hip, 1, STREAM) == 2                                                # p_choose hook lives in the packed chain
    assert _family(hip, 5, UNIT_BEAM) < 0                                              # no v1 unit decoder in a v2 model
    assert _family(hip, 0, GREEDY) < 0 and _family(hip, 1, 9) < 0


def test_dispatch_matrix_v1_unit_decoder():
    from seamless_communication_amd.config import tiny_v1_config
    from seamless_communication_amd import synthetic as syn
    from seamless_communication_amd.runtime import HipS2STModel

    cfg = tiny_v1_config()
    hip = HipS2STModel(cfg, syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED), None, device=0)
    assert [_family(hip, r, UNIT_BEAM) for r in (5, 64)] == [3, 3] and _family(hip, 80, UNIT_BEAM) == 4
    assert _family(hip, 5, BEAM) == 3 and _family(hip, 1, STREAM) < 0
    hip.close()
