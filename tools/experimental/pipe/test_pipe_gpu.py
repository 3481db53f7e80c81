"""GPU parity tests (-m gpu) of the pipelined strip kernel (algo = MLPG_HIP_ALGO_PIPE: three chunk wavefronts + one
chain wavefront per CU, level 1 of item s+1 under the level-2/3 chain of item s, halo sums handed over through LDS).
It is the strip scheme on another schedule, so the strip kernel's whole test module is run against it: every test of
tests/test_strip_gpu.py with ALGO_STRIP rebound to ALGO_PIPE (window sets other than three windows fall through to the
strip kernel itself, as the C ABI documents)."""
import pytest

import test_strip_gpu as S

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _pipe_instead_of_strip(monkeypatch):
    from nnmnkwii_amd import _hip
    monkeypatch.setattr(_hip, "ALGO_STRIP", _hip.ALGO_PIPE)


test_pipe_all_lengths = S.test_strip_all_lengths
test_pipe_static_dims = S.test_strip_static_dims
test_pipe_config2_utterances = S.test_strip_config2_utterances
test_pipe_not_pd_status = S.test_strip_not_pd_status
test_pipe_not_pd_far_from_strip0 = S.test_strip_not_pd_far_from_strip0
test_pipe_ill_conditioned = S.test_strip_ill_conditioned
test_pipe_long_range_coupling_falls_back_to_full_sweep = S.test_strip_long_range_coupling_falls_back_to_full_sweep
test_pipe_full_size_and_repeatability = S.test_strip_full_size_and_repeatability
test_pipe_two_streams_concurrently = S.test_strip_two_streams_concurrently
test_pipe_tight_dynamic_variances_every_window_rejected = S.test_strip_tight_dynamic_variances_every_window_rejected
test_pipe_kernel_inside_a_hip_graph = S.test_strip_kernel_inside_a_hip_graph
test_pipe_whole_utterance_route_many_groups_long_utterances = S.test_strip_whole_utterance_route_many_groups_long_utterances
test_pipe_whole_utterance_route_while_another_stream_holds_cus = S.test_strip_whole_utterance_route_while_another_stream_holds_cus


def test_pipe_really_runs_the_pipelined_kernel():
    """Guard against a silent fall-through: on a three-window problem ALGO_PIPE and ALGO_STRIP are different kernels with
    different strip lengths (48 vs 64 frames), so their float64 results differ in the last bits while both match the
    oracle -- and the halo variant of the level 1 must not change what the oracle says."""
    import numpy as np
    import torch
    from cases import WINDOW_SETS
    from nnmnkwii_amd import _hip
    from oracle import mlpg as O
    windows = WINDOW_SETS["std3"]
    rng = np.random.RandomState(3)
    B, T, sd = 4, 700, 60
    m = rng.randn(B, T, 3 * sd)
    v = rng.rand(B, T, 3 * sd) + 0.1
    mg, vg = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda()
    a, sa = _hip.forward(mg, vg, windows, algo=3)          # the strip kernel (literal: ALGO_STRIP is rebound here)
    b, sb = _hip.forward(mg, vg, windows, algo=_hip.ALGO_PIPE)
    assert int(sa.abs().max()) == 0 and int(sb.abs().max()) == 0
    yo, _, rc = O.mlpg_batch(m, v, windows)
    assert rc == 0
    scale = np.abs(yo).max()
    assert np.abs(a.cpu().numpy() - yo).max() <= 1e-9 * scale and np.abs(b.cpu().numpy() - yo).max() <= 1e-9 * scale
    assert not torch.equal(a, b)
