"""CPU: size-independent properties of the oracle (SURVEY.md §8c asks for property checks next to the golden vectors):
they hold for any input, so hypothesis can throw ragged / extreme cases at them."""
import numpy as np
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import nerfacto_oracle as orc

SETTINGS = dict(max_examples=25, deadline=None)


@settings(**SETTINGS)
@given(st.integers(1, 40), st.integers(1, 70), st.integers(0, 2**31 - 1))
def test_weights_are_a_sub_probability_and_order_preserving(n, s, seed):
    rs = np.random.RandomState(seed)
    t = np.sort(rs.uniform(0.05, 30.0, (n, s + 1)).astype(np.float32), axis=-1)
    dens = (np.exp(rs.standard_normal((n, s)) * 3) * rs.randint(0, 2, (n, s))).astype(np.float32)  # zeros and huge values
    w = orc.weights_from_density(torch.from_numpy(t), torch.from_numpy(dens))
    assert bool((w >= 0).all()) and float(w.sum(-1).max()) <= 1.0 + 1e-5
    assert bool((w[torch.from_numpy(dens) == 0] == 0).all())  # empty space contributes nothing


@settings(**SETTINGS)
@given(st.integers(1, 20), st.integers(2, 64), st.integers(1, 48), st.integers(0, 2**31 - 1), st.booleans())
def test_pdf_resample_stays_sorted_inside_the_support(n, s_prev, s_new, seed, training):
    rs = np.random.RandomState(seed)
    nears, fars = torch.full((n, 1), 0.05), torch.full((n, 1), 1000.0)
    s_bins, _ = orc.piecewise_bins(nears, fars, s_prev, torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)))
    w = torch.from_numpy((rs.uniform(0, 1, (n, s_prev)) ** 4 * rs.randint(0, 2, (n, 1))).astype(np.float32))  # some all-zero rays
    jit = torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)) if training else None
    s_new_bins, t_new, inds = orc.pdf_resample(s_bins, w, s_new, jit, nears, fars)
    assert s_new_bins.shape == (n, s_new + 1) and inds.dtype == torch.int64
    assert bool((s_new_bins[:, 1:] >= s_new_bins[:, :-1]).all()), "new bin edges are sorted"
    assert float(s_new_bins.min()) >= float(s_bins.min()) - 1e-6 and float(s_new_bins.max()) <= float(s_bins.max()) + 1e-6
    assert bool((t_new[:, 1:] >= t_new[:, :-1]).all()) and bool((inds >= 0).all()) and bool((inds <= s_prev + 1).all())


@settings(**SETTINGS)
@given(st.integers(1, 200), st.integers(0, 2**31 - 1), st.floats(0.01, 50.0))
def test_scene_contraction_maps_into_the_radius_2_cube_and_fixes_the_unit_cube(m, seed, scale):
    rs = np.random.RandomState(seed)
    x = torch.from_numpy((rs.standard_normal((m, 3)) * scale).astype(np.float32))
    y = orc.contract_linf(x)
    assert float(y.abs().max()) <= 2.0 + 1e-6
    inside = x.abs().max(-1).values < 1
    assert torch.equal(y[inside], x[inside])
    # direction (in the L-inf sense) is kept: the contracted point is a positive multiple of the input
    outside = ~inside
    if bool(outside.any()):
        ratio = y[outside] / x[outside]
        assert bool((ratio > 0).all()) and float((ratio.max(-1).values - ratio.min(-1).values).abs().max()) < 1e-4


@settings(**SETTINGS)
@given(st.integers(1, 6), st.integers(4, 12), st.integers(1, 300), st.integers(0, 2**31 - 1))
def test_hash_indices_stay_inside_their_level(num_levels, log2_t, m, seed):
    rs = np.random.RandomState(seed)
    T = 2**log2_t
    ix, iy, iz = (rs.randint(0, 4096, m) for _ in range(3))
    for lvl in range(num_levels):
        idx = orc.hash_corner_index(ix, iy, iz, lvl, T)
        assert idx.min() >= lvl * T and idx.max() < (lvl + 1) * T
        # int64 arithmetic modulo T (the reference, encodings.py:398-415) == the uint32 wrap-around restatement
        ref = ((ix.astype(object) * 1) ^ (iy.astype(object) * 2654435761) ^ (iz.astype(object) * 805459861))
        ref = np.array([int(v) % T for v in ref]) + lvl * T
        assert np.array_equal(idx, ref)


@settings(**SETTINGS)
@given(st.integers(1, 12), st.integers(2, 40), st.integers(0, 2**31 - 1))
def test_interlevel_loss_vanishes_when_the_proposal_histogram_bounds_the_fine_one(n, s, seed):
    """lossfun_outer is an upper-bound penalty (losses.py:85-102): a proposal histogram equal to the fine one costs 0,
    and the loss never goes negative."""
    rs = np.random.RandomState(seed)
    bins = torch.from_numpy(np.sort(rs.uniform(0, 1, (n, s + 1)).astype(np.float32), axis=-1))
    w = torch.from_numpy(rs.dirichlet(np.ones(s), n).astype(np.float32))
    same = orc.interlevel_loss([w, w], [bins, bins])
    assert abs(float(same)) < 1e-6
    other = torch.from_numpy(rs.dirichlet(np.ones(s), n).astype(np.float32))
    assert float(orc.interlevel_loss([other, w], [bins, bins])) >= 0.0
    assert float(orc.distortion_loss(w, bins)) >= 0.0
