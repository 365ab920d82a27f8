"""N>1 path on CPU: world_size-2 gloo processes shard a batch and all-gather logits in rank-major order (the harness of
tests/dist_harness.py with a stand-in operator; tests/test_dist_gpu.py runs the same harness with the real KeyedModel)."""
import pytest

from keynet_amd import dist as kdist
import dist_harness


@pytest.mark.parametrize('n', [8, 7])
def test_sharded_forward_equals_single_process(n):
    res = dist_harness.run('standin', n)
    assert [r[1] for r in res] == [True, True]
    assert res[0][2][0] == 0 and res[0][2][1] == res[1][2][0] and res[1][2][1] == n
    assert all(r[3] == (n, 5) for r in res)


def test_sharded_forward_on_eight_ranks_ragged():
    """BASELINE configs[4] is an 8-rank job: eight gloo processes, a ragged total of 61 rows (shards of 8 and 7), gathered logits bit-equal to the
    single-process forward on every rank, rank-major, one all_gather_into_tensor per forward (tests/test_dist_gpu.py: the same with the real kernels)."""
    n = 61
    res = dist_harness.run('standin', n, world_size=8)
    assert [r[0] for r in res] == list(range(8)) and all(r[1] for r in res)
    bounds = [r[2] for r in res]
    assert bounds[0][0] == 0 and bounds[-1][1] == n and all(bounds[k][1] == bounds[k + 1][0] for k in range(7))
    assert sorted(set(hi - lo for (lo, hi) in bounds)) == [7, 8]
    assert all(r[3] == (n, 5) and len(r[6]) == 1 for r in res)


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 256, 2048, 2049):
        for ws in (1, 2, 4, 8):
            b = [kdist.shard_bounds(n, r, ws) for r in range(ws)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for (lo, hi) in b]
            assert max(sizes) - min(sizes) <= 1


def test_calibration_decisions_are_made_collective():
    """Ranks that calibrated on different batches agree after KeyedModel.sync_contract: a layer every rank decided alike keeps its contract, any disagreement ends in
    the reference's order everywhere (an exact decision anywhere wins; two DIFFERENT re-ordering contracts -- bf16x3 here, the f32 matrix cores there -- are not
    ordered: neither rank's calibration record covers the other's kernel), only the ranks that were elsewhere change, and a second round is a no-op."""
    res = dist_harness.run_contract('host')
    ((_, ch0, st0, again0, s1_0, s2_0), (_, ch1, st1, again1, s1_1, s2_1)) = res
    assert st0 == st1 == {'conv1': True, 'pool1': True, 'conv2': True, 'pool2': True, 'fc1': True}
    assert sorted(ch0) == ['conv1', 'conv2'] and ch1 == ['conv2']
    assert again0 == [] and again1 == []
    assert not s1_0 and not s1_1 and not s2_0 and not s2_1      # both run in the reference's order everywhere: nothing left to screen


def test_split_decisions_are_made_collective():
    """The 'split' contract (a filled-in conv applied as spatial mixing then channel mixing) in KeyedModel.sync_contract: a layer one rank runs split and another
    fused ends in the reference's order on both (the rank that accepted 'split' never measured the fused kernel on its batch, and the other way round: round-5
    advisor finding); a layer both run split stays split and screened against each rank's own calibration."""
    res = dist_harness.run_contract('host_split')
    ((_, ch0, st0, again0, s1_0, s2_0), (_, ch1, st1, again1, s1_1, s2_1)) = res
    assert st0 == st1 == {'conv1': True, 'pool1': True, 'conv2': 'split', 'pool2': True, 'fc1': True}
    assert ch0 == ['conv1'] and ch1 == ['conv1'] and again0 == [] and again1 == []
    assert not s1_0 and not s1_1 and s2_0 and s2_1
