"""Case table shared by make_golden.py (which runs the REFERENCE keygen on it) and tests/test_keygen_families.py
(which runs keynet_amd.system.keygen on it).  Pure data."""

KEYGEN_CASES = [
    # (name, shape, kwargs) -- every key family of keynet/system.py:317-469 on small shapes
    ('perm_global', (2, 8, 8), dict(global_geometric='permutation', local_geometric='identity', global_photometric='identity', local_photometric='identity')),
    ('perm_local_b4', (3, 8, 8), dict(global_geometric='identity', local_geometric='permutation', global_photometric='identity', local_photometric='identity', blocksize=4)),
    ('block_order_b4', (2, 8, 8), dict(global_geometric='identity', local_geometric='identity', global_photometric='identity', local_photometric='identity', memoryorder='block', blocksize=4)),
    ('hier_perm', (2, 16, 16), dict(global_geometric='hierarchical_permutation', local_geometric='identity', global_photometric='identity', local_photometric='identity',
                                    hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0, 1))),
    ('hier_rot', (1, 16, 16), dict(global_geometric='hierarchical_rotation', local_geometric='identity', global_photometric='identity', local_photometric='identity',
                                   hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0))),
    ('hier_perm_block', (2, 16, 16), dict(global_geometric='hierarchical_permutation', local_geometric='identity', global_photometric='identity', local_photometric='identity',
                                          hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0, 1), memoryorder='block', blocksize=4)),
    ('givens_global', (1, 6, 6), dict(global_geometric='givens_orthogonal', local_geometric='identity', global_photometric='identity', local_photometric='identity', alpha=5)),
    ('givens_local_b4', (2, 8, 8), dict(global_geometric='identity', local_geometric='givens_orthogonal', global_photometric='identity', local_photometric='identity', alpha=4, blocksize=4)),
    ('dstoch_local_b4', (2, 8, 8), dict(global_geometric='identity', local_geometric='doubly_stochastic', global_photometric='identity', local_photometric='identity', alpha=2, blocksize=4)),
    ('gain_global', (2, 4, 4), dict(global_geometric='identity', local_geometric='identity', global_photometric='uniform_random_gain', local_photometric='identity', beta=1.0)),
    ('bias_global', (2, 4, 4), dict(global_geometric='identity', local_geometric='identity', global_photometric='uniform_random_bias', local_photometric='identity', gamma=1.0)),
    ('affine_global', (2, 4, 4), dict(global_geometric='identity', local_geometric='identity', global_photometric='uniform_random_affine', local_photometric='identity', beta=1.0, gamma=1.0)),
    ('linear_bias', (2, 4, 4), dict(global_geometric='identity', local_geometric='identity', global_photometric='linear_bias', local_photometric='identity', gamma=2.0)),
    ('blockwise_bias', (2, 8, 8), dict(global_geometric='identity', local_geometric='identity', global_photometric='blockwise_constant_bias', local_photometric='identity', gamma=2.0, blocksize=4)),
    ('gain_local_b4', (2, 8, 8), dict(global_geometric='identity', local_geometric='identity', global_photometric='identity', local_photometric='uniform_random_gain', beta=1.0, blocksize=4)),
    ('bias_local_b4', (2, 8, 8), dict(global_geometric='identity', local_geometric='identity', global_photometric='identity', local_photometric='uniform_random_bias', gamma=1.0, blocksize=4)),
    ('affine_local_b4', (2, 8, 8), dict(global_geometric='identity', local_geometric='identity', global_photometric='identity', local_photometric='uniform_random_affine', beta=0.1, gamma=100.0, blocksize=4)),
    ('orthogonal_family', (2, 16, 16), dict(global_geometric='hierarchical_permutation', hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0, 1), global_photometric='identity',
                                            local_geometric='givens_orthogonal', alpha=4, blocksize=4, local_photometric='uniform_random_affine', beta=0.1, gamma=100.0,
                                            memoryorder='block', tileshape=(4, 4))),
    ('fc_vector', (10, 1, 1), dict(global_geometric='identity', local_geometric='givens_orthogonal', global_photometric='identity', local_photometric='uniform_random_affine',
                                   alpha=4, beta=0.1, gamma=100.0, blocksize=4, memoryorder='block')),
    # banded doubly-stochastic keys with MORE than two diagonals: the Sinkhorn normalisation sums 3..5 entries per row/column,
    # so the summation order of the normaliser is pinned too (two-entry sums commute and cannot tell)
    ('dstoch_local_b4_k3', (1, 8, 8), dict(global_geometric='identity', local_geometric='doubly_stochastic', global_photometric='identity', local_photometric='identity', alpha=3, blocksize=4)),
    ('dstoch_local_b4_k5', (2, 8, 8), dict(global_geometric='identity', local_geometric='doubly_stochastic', global_photometric='identity', local_photometric='identity', alpha=5, blocksize=4)),
    ('givens_global_k40', (1, 6, 6), dict(global_geometric='givens_orthogonal', local_geometric='identity', global_photometric='identity', local_photometric='identity', alpha=40)),
    ('hier_perm_3lvl', (1, 32, 32), dict(global_geometric='hierarchical_permutation', local_geometric='identity', global_photometric='identity', local_photometric='identity',
                                         hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0, 1, 2))),
    ('hier_rot_2lvl', (2, 16, 16), dict(global_geometric='hierarchical_rotation', local_geometric='identity', global_photometric='identity', local_photometric='identity',
                                        hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0, 1))),
    ('ragged_blocksize', (1, 14, 14), dict(global_geometric='identity', local_geometric='permutation', global_photometric='identity', local_photometric='identity', blocksize=4)),
]

# keyword arguments of test/test_keynet.py:116-129 (test_vgg16_stochastic) scaled to MiniNet (2,16,16): tile 4, block 4, two hierarchy levels.
# make_golden.py f8 hands them to the REFERENCE's keynet.system.Keynet, the tests to keynet_amd.system.Keynet.
STOCHASTIC_KW = dict(tileshape=(4, 4), global_geometric='hierarchical_permutation', hierarchical_blockshape=(2, 2), hierarchical_permute_at_level=(0, 1),
                     local_geometric='doubly_stochastic', alpha=2.0, blocksize=4, local_photometric='uniform_random_affine', beta=1.0, gamma=1.0)
