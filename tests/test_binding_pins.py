"""The binding-behaviour pins of the reference (tests/test_swig.py:9-111) restated against gridpp_amd: element types of
results, list / tuple / any-dtype inputs, wrong ndim raises, zero-size dimensions.  No GPU needed (the helpers only go through
the conversion layer every entry point shares); the same element-type contract is asserted on real results in the GPU tests."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def gridpp():
    import gridpp_amd
    return gridpp_amd


def test_int_output(gridpp):                               # test_swig.py:9-15
    assert type(gridpp.test_ivec_output()[0]) == np.int32
    assert type(gridpp.test_ivec2_output()[0][0]) == np.int32


def test_float_output(gridpp):                             # :17-23
    assert type(gridpp.test_vec_output()[0]) == np.float32
    assert type(gridpp.test_vec2_output()[0][0]) == np.float32
    assert type(gridpp.test_vec3_output()[0][0][0]) == np.float32


def test_vec_input(gridpp):                                # :25-35
    ar = [1, 2, 3]
    for func in [gridpp.test_vec_input, gridpp.test_ivec_input]:
        assert func(ar) == 6
        assert func((1, 2, 3)) == 6
        assert func(np.array(ar)) == 6
        for dt in ("float32", "float64", "int32"):
            assert func(np.array(ar).astype(dt)) == 6


def test_vec2_vec3_input(gridpp):                          # :37-51
    ar = [[1, 2], [2, 3], [3, 4]]
    ar3 = [[[1, 2], [2, 3]], [[2, 3], [3, 4]], [[3, 4], [4, 5]]]
    for a, f, want in ((ar, gridpp.test_vec2_input, 15), (ar3, gridpp.test_vec3_input, 36)):
        assert f(a) == want
        assert f(np.array(a)) == want
        for dt in ("float32", "float64", "int32"):
            assert f(np.array(a).astype(dt)) == want


def test_argout(gridpp):                                   # :53-60
    n, distances = gridpp.test_vec_argout()
    assert len(distances) == 10
    n, distances = gridpp.test_vec2_argout()
    assert distances.shape == (10, 10)


def test_invalid_dimension_error(gridpp):                  # :62-72
    for func in [gridpp.test_vec2_input, gridpp.test_vec3_input]:
        with pytest.raises(Exception):
            func(np.zeros([5]))
    for func in [gridpp.test_vec_input, gridpp.test_vec3_input]:
        with pytest.raises(Exception):
            func(np.zeros([5, 2]))
    for func in [gridpp.test_vec_input, gridpp.test_vec2_input]:
        with pytest.raises(Exception):
            func(np.zeros([5, 2, 3]))


def test_outputs(gridpp):                                  # :74-84
    ar = [-1, -1, -1]
    np.testing.assert_array_equal(gridpp.test_vec_output(), ar)
    np.testing.assert_array_equal(gridpp.test_vec2_output(), [ar, ar, ar])
    np.testing.assert_array_equal(gridpp.test_vec3_output(), [[ar, ar, ar]] * 3)
    np.testing.assert_array_equal(gridpp.test_ivec_output(), ar)
    np.testing.assert_array_equal(gridpp.test_ivec2_output(), [ar, ar, ar])
    np.testing.assert_array_equal(gridpp.test_ivec3_output(), [[ar, ar, ar]] * 3)
    gridpp.test_array([1, 2, 3])                           # :86-89


def test_zero_size_dimension(gridpp):                      # :91-107
    assert gridpp.test_vec_input(np.zeros([0])) == 0
    assert gridpp.test_vec_input([]) == 0
    for shape in ([3, 0], [0, 3], [0, 0]):
        assert gridpp.test_vec2_input(np.zeros(shape)) == 0
    for shape in ([3, 3, 0], [3, 0, 3], [0, 3, 3], [3, 0, 0], [0, 3, 0], [0, 0, 3]):
        assert gridpp.test_vec3_input(np.zeros(shape)) == 0


def test_not_implemented_exception(gridpp):                # :109-111
    with pytest.raises(RuntimeError):
        gridpp.test_not_implemented_exception()


def test_entry_points_share_the_conversion_layer(gridpp):
    """The same rules hold at a real entry point without touching the GPU: ndim check and empty input of neighbourhood."""
    with pytest.raises(Exception):
        gridpp.neighbourhood(np.zeros([5]), 1, gridpp.Mean)
    out = gridpp.neighbourhood(np.zeros([0, 0]), 1, gridpp.Mean)
    assert np.asarray(out).size == 0
